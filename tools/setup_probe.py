#!/usr/bin/env python3
"""Wall time of set_target / set_source (upload, centring, records, grid build, source order) at 1M points."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import icp_amd
from icp_amd import synth
P, Q = synth.make_pair(1000000, 1000000, seed=5)
rng = np.random.RandomState(1)
u = rng.uniform(-1, 1, (1000000, 2))
S = np.column_stack([u[:, 0], u[:, 1], 0.1 * np.sin(5 * u[:, 0]) * np.cos(4 * u[:, 1])])
for name, T in (("cube", Q), ("sheet", S)):
    for search in ("grid", "f32"):
        e = icp_amd.Engine(search=search)
        e.set_target(T); e.set_source(P)          # warm
        t0 = time.perf_counter(); e.set_target(T); t1 = time.perf_counter(); e.set_source(P); t2 = time.perf_counter()
        e.close()
        print("%-5s %-4s: set_target %.2f ms, set_source %.2f ms" % (name, search, (t1 - t0) * 1e3, (t2 - t1) * 1e3), flush=True)
