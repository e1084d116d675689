#!/usr/bin/env python3
"""Per-iteration time of the engine on small / medium clouds (launch-overhead regime)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import icp_amd
from icp_amd import synth
shapes = [(30, 30), (1000, 1000), (2903, 2903), (20000, 20000), (48485, 48485), (100000, 100000), (300000, 300000),
          (125952, 1000000)]
for n, m in shapes:
    P, Q = synth.make_pair(n, m, seed=5)
    e = icp_amd.Engine()
    e.set_target(Q); e.set_source(P)
    e.iterate(3); e.sync()
    it = 50 if n * m < 2e10 else 10
    ms = min(e.iterate_timed(it) for _ in range(3)) / it
    g = e.search_geometry()
    e.close()
    print("%7d x %7d: %.1f us/iteration  (%.2e evals/s)  %s" % (n, m, ms * 1e3, n * float(m) / (ms * 1e-3), g), flush=True)
