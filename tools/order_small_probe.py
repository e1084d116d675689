#!/usr/bin/env python3
"""Is the source's spatial order (grid_order, computed at set_source) worth its set-up time on small clouds?  Per size: wall
time of set_source and of 20 iterations from a fresh source (first, un-seeded search included), with and without the order."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import icp_amd
from icp_amd import synth
for n, m in ((2903, 2903), (20000, 20000), (48485, 48485), (100000, 100000), (200000, 200000), (300000, 300000), (30000, 1000000),
             (125952, 1000000)):
    P, Q = synth.make_pair(n, m, seed=5)
    e = icp_amd.Engine(search="grid")
    e.set_target(Q); e.set_source(P); e.iterate(20); e.sync()
    ts, ti = [], []
    for rep in range(5):
        t0 = time.perf_counter(); e.set_source(P); t1 = time.perf_counter(); e.iterate(20); e.sync(); t2 = time.perf_counter()
        ts.append((t1 - t0) * 1e3); ti.append((t2 - t1) * 1e3)
    e.close()
    print("order=%s %7d x %7d: set_source %.3f ms, 20 iterations %.3f ms, both %.3f ms" %
          (os.environ.get("ICP_AMD_GRID_ORDER", "auto"), n, m, min(ts), min(ti), min(a + b for a, b in zip(ts, ti))), flush=True)
