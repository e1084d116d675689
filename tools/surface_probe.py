#!/usr/bin/env python3
"""Grid search on surface-like clouds (points on a sphere / a bumpy sheet), where most grid cells are empty."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import icp_amd
from icp_amd import synth

def surface(n, kind, seed):
    rng = np.random.RandomState(seed)
    if kind == "sphere":
        v = rng.standard_normal((n, 3))
        return v / np.linalg.norm(v, axis=1)[:, None] * np.array([1.0, 0.7, 0.5])
    u = rng.uniform(-1, 1, (n, 2))
    return np.column_stack([u[:, 0], u[:, 1], 0.1 * np.sin(5 * u[:, 0]) * np.cos(4 * u[:, 1])])

R0 = synth.rotation(0.05) if hasattr(synth, "rotation") else None
for kind in ("sphere", "sheet"):
    for n in (48485, 1000000):
        Q = surface(n, kind, 1)
        rng = np.random.RandomState(2)
        c, s = np.cos(0.05), np.sin(0.05)
        R = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1.0]])
        P = Q[rng.permutation(n)] @ R.T + np.array([0.01, -0.02, 0.015]) + 1e-4 * rng.standard_normal((n, 3))
        e = icp_amd.Engine(search="grid", kernel_timing=True)
        e.set_target(Q); e.set_source(P)
        e.iterate(1); e.sync()
        first = e.kernel_ms()[0]
        e.iterate(4); e.sync(); e.kernel_ms()
        ms = min(e.iterate_timed(20) for _ in range(3)) / 20
        st = e.grid_stats()
        e.close()
        print("%-6s %7d: %.1f us/iteration (first search %.3f ms) %s" % (kind, n, ms * 1e3, first, st), flush=True)
