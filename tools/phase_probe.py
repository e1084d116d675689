#!/usr/bin/env python3
"""Where the end-to-end time of a one-shot registration goes (1M x 1M, 20 iterations, grid search): wall time of each
C-ABI call of a fresh engine, second registration of the process (the HIP context and code objects are warm)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import icp_amd
from icp_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
P, Q = synth.make_pair(n, n, seed=5)
for rep in range(3):
    t = [time.perf_counter()]
    e = icp_amd.Engine(); t.append(time.perf_counter())
    e.set_target(Q); t.append(time.perf_counter())
    e.set_source(P); t.append(time.perf_counter())
    e.iterate(iters); t.append(time.perf_counter())
    e.sync(); t.append(time.perf_counter())
    out = e.get_source(); t.append(time.perf_counter())
    e.close(); t.append(time.perf_counter())
    names = ("create", "set_target", "set_source", "iterate (enqueue)", "sync", "get_source", "destroy")
    print("run %d: " % rep + ", ".join("%s %.2f" % (k, (b - a) * 1e3) for k, a, b in zip(names, t, t[1:])) + "  | total %.2f ms" % ((t[-1] - t[0]) * 1e3), flush=True)
t0 = time.perf_counter(); icp_amd.icp_gpu_optimized(P, Q, iters); print("icp_amd_run: %.2f ms" % ((time.perf_counter() - t0) * 1e3))
