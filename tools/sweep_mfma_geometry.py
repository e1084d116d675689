#!/usr/bin/env python3
"""Source groups per wave (ICP_AMD_MFMA_G) x target splits of the matrix-pipe brute-force search, at the 8-GPU shard shape,
the 4- and 2-GPU shard shapes and the full 1M x 1M: per-iteration time in microseconds."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import icp_amd
from icp_amd import synth
shapes = ((125952, 1000000, (12, 18, 24, 36, 48, 72)), (250880, 1000000, (9, 12, 18, 24, 36)), (500736, 1000000, (6, 9, 12, 18, 24)),
          (1000000, 1000000, (5, 9, 12, 17, 24)))
for n, m, splits in shapes:
    P, Q = synth.make_pair(n, m, seed=5)
    e = icp_amd.Engine(search="f32"); e.set_target(Q); e.set_source(P); e.iterate(3); e.sync()
    us = min(e.iterate_timed(5) for _ in range(3)) / 5 * 1e3
    g = e.search_geometry(); e.close()
    print("%d x %d default: q_splits %d, workgroups %d: %.0f us" % (n, m, g["q_splits"], g["workgroups"], us), flush=True)
    for G in (2, 4):
        line = []
        for sp in splits:
            os.environ.update(ICP_AMD_MFMA_G=str(G), ICP_AMD_QSPLITS=str(sp))
            e = icp_amd.Engine(search="f32")
            e.set_target(Q); e.set_source(P)
            e.iterate(3); e.sync()
            us = min(e.iterate_timed(5) for _ in range(3)) / 5 * 1e3
            g = e.search_geometry()
            e.close()
            line.append("sp=%d(wg %d) %.0f" % (g["q_splits"], g["workgroups"], us))
        os.environ.pop("ICP_AMD_MFMA_G"); os.environ.pop("ICP_AMD_QSPLITS")
        print("%d x %d G=%d: %s" % (n, m, G, " | ".join(line)), flush=True)
