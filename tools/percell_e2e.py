#!/usr/bin/env python3
"""Targets per grid cell against the whole run (set-up + un-seeded first pass + 20 iterations, host buffers in/out)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import icp_amd
from icp_amd import synth
sys.path.insert(0, os.path.join(ROOT, "tools"))
from percell_sweep import sheet
icp_amd.icp_gpu_optimized(*synth.make_pair(2000, 2000), 2)
for name, n in (("cube", 48485), ("cube", 1000000), ("sheet", 1000000)):
    P, Q = synth.make_pair(n, n, seed=5) if name == "cube" else sheet(n)
    line = []
    for pc in ("0.5", "1", "1.5", "2", "4"):
        os.environ["ICP_AMD_GRID_PER_CELL"] = pc
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter(); icp_amd.icp_gpu_optimized(P, Q, 20); best = min(best, time.perf_counter() - t0)
        line.append("%s: %.2f ms" % (pc, best * 1e3))
    print("%-5s %7d x 20 iterations end to end, by targets per cell  %s" % (name, n, " | ".join(line)), flush=True)
