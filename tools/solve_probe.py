#!/usr/bin/env python3
"""icp_amd_solve called back to back (under rocprofv3 --kernel-trace: solve_kernel with one partial record and nothing else
on the GPU in between -- the kernel's time without a cold instruction cache or a long partial sum)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import icp_amd
rng = np.random.RandomState(1)
A = rng.standard_normal((3, 3)) + 3 * np.eye(3)
for _ in range(300):
    icp_amd.solve(A, np.zeros(3), np.ones(3))
print("done")
