#!/usr/bin/env python3
"""Source-groups x target-splits sweep of the fp16 MFMA filter kernel (brute-force search, F32 variant)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import icp_amd
from icp_amd import synth
n, m = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1000000, 1000000)
splits = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [4, 6, 9, 12, 18, 27, 40]
P, Q = synth.make_pair(n, m, seed=5)
for g in (2, 4):
    line = []
    for sp in splits:
        os.environ.update(ICP_AMD_MFMA="16", ICP_AMD_MFMA_G=str(g), ICP_AMD_QSPLITS=str(sp))
        e = icp_amd.Engine(search="f32")
        e.set_target(Q); e.set_source(P)
        e.iterate(3); e.sync()
        us = min(e.iterate_timed(5) for _ in range(2)) / 5 * 1e3
        geo = e.search_geometry()
        e.close()
        line.append("sp=%d(wg %d) %.0f" % (geo["q_splits"], geo["workgroups"], us))
    print("%d x %d groups=%d: %s" % (n, m, g, " | ".join(line)), flush=True)
