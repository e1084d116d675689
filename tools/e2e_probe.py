#!/usr/bin/env python3
"""End-to-end wall time of icp_amd_run (host buffers in, host buffers out: PCIe, set-up and grid build included)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import icp_amd
from icp_amd import synth
icp_amd.icp_gpu_optimized(*synth.make_pair(1000, 1000), 2)   # context creation is not part of the measurement
for n, iters in ((2903, 20), (48485, 20), (100000, 30), (1000000, 20)):
    P, Q = synth.make_pair(n, n, seed=5)
    for search in ("auto", "f32"):
        if search == "f32" and n > 100000:
            continue
        os.environ["ICP_AMD_SEARCH"] = search
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            out = icp_amd.icp_gpu_optimized(P, Q, iters)
            best = min(best, time.perf_counter() - t0)
        print("%7d x %7d, %2d iterations, search=%-4s: %.2f ms end to end" % (n, n, iters, search, best * 1e3), flush=True)
