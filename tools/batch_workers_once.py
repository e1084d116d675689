#!/usr/bin/env python3
"""One size, a list of worker counts (each run_batch call prints its throughput): python tools/batch_workers_once.py 20000 128 4 8 8"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import icp_amd
from icp_amd import synth
n, b = int(sys.argv[1]), int(sys.argv[2])
pairs = [synth.make_pair(n, n, seed=500 + k) for k in range(8)]
P = np.stack([pairs[k % 8][0] for k in range(b)]); Q = np.stack([pairs[k % 8][1] for k in range(b)])
ref = None
for w in sys.argv[3:]:
    os.environ["ICP_AMD_BATCH_WORKERS"] = w
    t0 = time.perf_counter(); d = icp_amd.run_batch(P, Q, 20); dt = time.perf_counter() - t0
    if ref is None: ref = d["P_out"].copy()
    print("%d x (%d x %d), %s workers: %.1f ms = %.0f registrations/s, same results %s" % (b, n, n, w, dt * 1e3, b / dt, bool(np.array_equal(ref, d["P_out"]))), flush=True)
