#!/usr/bin/env python3
"""Timing probe for the multi-registration kernel: fixed cost per iteration vs search cost."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import icp_amd
from icp_amd import synth
for batch, n, m in ((4096, 1000, 1000), (4096, 64, 64), (256, 1000, 1000), (768, 1000, 1000), (4096, 256, 256)):
    P, Q = synth.make_batch(min(batch, 32), max(n, m))
    reps = (batch + len(P) - 1) // len(P)
    P = np.tile(P[:, :n], (reps, 1, 1))[:batch]; Q = np.tile(Q[:, :m], (reps, 1, 1))[:batch]
    b = icp_amd.Batch(batch, n, m)
    b.upload(P, Q)
    b.run(2)
    for iters in (0, 1, 5, 20):
        ms = min(b.run(iters, timed=True) for _ in range(3))
        print("batch=%d n=%d m=%d iters=%2d: %.3f ms" % (batch, n, m, iters, ms), flush=True)
    b.close()
