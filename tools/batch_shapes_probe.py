#!/usr/bin/env python3
"""Multi-registration kernel over cloud sizes: launch time, registrations per second and distance evaluations per second for
20 iterations, default search and ICP_AMD_BATCH_SEEDED=0 (set by the caller).  GPU box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import icp_amd
from icp_amd import synth

print("ICP_AMD_BATCH_SEEDED =", os.environ.get("ICP_AMD_BATCH_SEEDED", "unset (seeded from 384 targets on)"), flush=True)
for B, n, m in ((16384, 256, 256), (8192, 512, 512), (4096, 1000, 1000), (4096, 1024, 2048), (2048, 2048, 2048), (1024, 2048, 4096), (256, 1000, 1000)):
    P, Q = synth.make_batch(32, max(n, m), seed=synth.SEED_Q)
    P, Q = np.tile(P[:, :n], (B // 32, 1, 1)), np.tile(Q[:, :m], (B // 32, 1, 1))
    b = icp_amd.Batch(B, n, m)
    b.upload(P, Q)
    for _ in range(2):
        b.run(20)
    ms = min(b.run(20, timed=True) for _ in range(3))
    b.close()
    print("%6d registrations of %4d x %4d: %8.3f ms  %9.0f registrations/s  %.2e evals/s" % (B, n, m, ms, B / ms * 1e3, B * n * m * 20.0 / ms * 1e3), flush=True)
