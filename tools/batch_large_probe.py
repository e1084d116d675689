#!/usr/bin/env python3
"""Throughput of icp_amd_run_batch on clouds beyond the batch kernel's limits (registrations through ordinary engines,
ICP_AMD_BATCH_WORKERS at a time): 128 x (20k x 20k) and 64 x (100k x 100k), 20 iterations each."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import icp_amd
from icp_amd import synth
for b, n in ((128, 20000), (64, 100000)):
    pairs = [synth.make_pair(n, n, seed=500 + k) for k in range(8)]
    P = np.stack([pairs[k % 8][0] for k in range(b)]); Q = np.stack([pairs[k % 8][1] for k in range(b)])
    icp_amd.run_batch(P[:8], Q[:8], 2)
    for w in (1, 2, 4, 8):
        os.environ["ICP_AMD_BATCH_WORKERS"] = str(w)
        t0 = time.perf_counter(); icp_amd.run_batch(P, Q, 20); dt = time.perf_counter() - t0
        print("%4d x (%6d x %6d) x 20 iterations, %2d workers: %.1f ms = %.0f registrations/s" % (b, n, n, w, dt * 1e3, b / dt), flush=True)
