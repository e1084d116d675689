#!/usr/bin/env python3
"""The first eight iterations one by one (ms), fresh source each repetition: 1M x 1M cube, 100k, cow-size."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import icp_amd
from icp_amd import synth
for n in (1000000, 100000, 2903):
    P, Q = synth.make_pair(n, n, seed=5)
    e = icp_amd.Engine(search="grid")
    e.set_target(Q)
    best = None
    for rep in range(4):
        e.set_source(P)
        its = [e.iterate_timed(1) for _ in range(8)]
        best = its if best is None else [min(a, b) for a, b in zip(best, its)]
    e.close()
    print("%7d rows, iterations 0..7 (ms): %s  sum %.3f" % (n, " ".join("%.4f" % x for x in best), sum(best)), flush=True)
