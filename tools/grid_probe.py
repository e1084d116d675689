#!/usr/bin/env python3
"""Per-iteration time of the engine with the grid search next to the brute-force default."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import icp_amd
from icp_amd import synth
shapes = [(2903, 2903), (20000, 20000), (48485, 48485), (100000, 100000), (1000000, 1000000), (125952, 1000000)]
for n, m in shapes:
    P, Q = synth.make_pair(n, m, seed=5)
    for search in ("grid", "f32"):
        if search == "f32" and n * m > 2e10:
            continue
        e = icp_amd.Engine(search=search, kernel_timing=True)
        e.set_target(Q); e.set_source(P)
        e.iterate(1); e.sync()
        first = e.kernel_ms()[0]
        e.iterate(2); e.sync()
        it = 50
        ms = min(e.iterate_timed(it) for _ in range(3)) / it
        k = e.kernel_ms()
        e.close()
        print("%7d x %7d %-4s: %.1f us/iteration  (search kernel first %.3f ms, avg %.3f ms)" % (n, m, search, ms * 1e3, first, k[0]), flush=True)
