#!/usr/bin/env python3
"""Sweep points-per-lane x target splits of the f32 search on small / medium clouds: us per full iteration."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import icp_amd
from icp_amd import synth
sizes = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1000,2903,20000,48485,100000,300000").split(",")]
for n in sizes:
    P, Q = synth.make_pair(n, n, seed=5)
    scan = (n + 63) // 64
    for ppl in (2, 4, 8):
        xb = -(-n // 2048) * 2048 // (256 * ppl)
        line = []
        for target_wgs in (256, 512, 1024, 2048, 4096, 8192, 16384):
            sp = max(1, min(scan, -(-target_wgs // xb)))
            os.environ.update(ICP_AMD_PPL=str(ppl), ICP_AMD_QSPLITS=str(sp))
            e = icp_amd.Engine()
            e.set_target(Q); e.set_source(P)
            e.iterate(3); e.sync()
            iters = 40 if n <= 100000 else 10
            us = min(e.iterate_timed(iters) for _ in range(3)) / iters * 1e3
            g = e.search_geometry()
            e.close()
            line.append("sp=%d(wg %d) %.0f" % (g["q_splits"], g["workgroups"], us))
        print("n=%d ppl=%d: %s" % (n, ppl, " | ".join(line)), flush=True)
