#!/usr/bin/env python3
"""Points-per-lane x target-split sweep of the brute-force F32 search on the 8-GPU shard shape (125952 x 1M)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import icp_amd
from icp_amd import synth
n, m = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (125952, 1000000)
P, Q = synth.make_pair(n, m, seed=5)
for ppl in (4, 8):
    line = []
    for sp in (24, 36, 48, 72, 96, 144, 200):
        os.environ.update(ICP_AMD_PPL=str(ppl), ICP_AMD_QSPLITS=str(sp))
        e = icp_amd.Engine(search="f32")
        e.set_target(Q); e.set_source(P)
        e.iterate(3); e.sync()
        us = min(e.iterate_timed(10) for _ in range(3)) / 10 * 1e3
        g = e.search_geometry()
        e.close()
        line.append("sp=%d(wg %d) %.0f" % (g["q_splits"], g["workgroups"], us))
    print("%d x %d ppl=%d: %s" % (n, m, ppl, " | ".join(line)), flush=True)
