#!/usr/bin/env python3
"""Multi-registration kernel: launch time against the iteration count (4096 registrations of 1000 x 1000 points), i.e. what
the first iteration (expanded form), the moving iterations and the converged ones cost.  GPU box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import icp_amd
from icp_amd import synth

B, n = 4096, int(sys.argv[1]) if len(sys.argv) > 1 else 1000
P, Q = synth.make_batch(64, n, seed=synth.SEED_Q)
P, Q = np.tile(P, (B // 64, 1, 1)), np.tile(Q, (B // 64, 1, 1))
b = icp_amd.Batch(B, n, n)
b.upload(P, Q)
prev = 0.0
for iters in (1, 2, 3, 4, 6, 8, 10, 15, 20, 30):
    for _ in range(2):
        b.run(iters)
    ms = min(b.run(iters, timed=True) for _ in range(3))
    print("iters %2d  %7.3f ms   +%6.3f ms per added iteration" % (iters, ms, (ms - prev) / (iters - getattr(b, "_pi", 0))), flush=True)
    prev, b._pi = ms, iters
b.close()
