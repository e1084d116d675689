#!/usr/bin/env python3
"""Grid iteration time from the start of a 1M x 1M registration, ten iterations per figure (ICP_AMD_GRID_ORDER_SWITCH sweep)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import icp_amd
from icp_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
P, Q = synth.make_pair(n, n)
e = icp_amd.Engine(search="grid")
e.set_target(Q); e.set_source(P)
e.iterate(3); e.sync()
ms = [e.iterate_timed(10) / 10 for _ in range(16)]
e.close()
print("switch=%s order=%s: %s  total %.2f ms" % (os.environ.get("ICP_AMD_GRID_ORDER_SWITCH", "default"), os.environ.get("ICP_AMD_GRID_ORDER", "1"),
                                   " ".join("%.4f" % m for m in ms), 10 * sum(ms)), flush=True)
