#!/usr/bin/env python3
"""Variant-against-variant fuzz at sizes the CPU oracle cannot follow (up to 300k points): grid vs F32 vs EXACT engines,
several seeded iterations, on volume-filling, surface-like, clustered and partially overlapping clouds."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import icp_amd

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t_end = time.time() + budget
t_say = time.time() + 30
cases = bad = 0
while time.time() < t_end:
    if time.time() > t_say:
        print("... %d cases, %d mismatches so far" % (cases, bad), flush=True)
        t_say = time.time() + 30
    n, m, iters = int(rng.randint(1000, 300000)), int(rng.randint(1000, 300000)), int(rng.randint(2, 13))
    kind = rng.randint(4)
    if kind == 0:
        Q = rng.uniform(-1, 1, (m, 3))
    elif kind == 1:
        v = rng.standard_normal((m, 3)); Q = v / np.linalg.norm(v, axis=1)[:, None] * rng.uniform(0.3, 2, 3)
    elif kind == 2:
        c = rng.uniform(-1, 1, (50, 3)); Q = c[rng.randint(0, 50, m)] + 0.02 * rng.standard_normal((m, 3))
    else:
        u = rng.uniform(-1, 1, (m, 2)); Q = np.column_stack([u, 0.2 * np.sin(4 * u[:, 0])])
    a = rng.uniform(0, 0.3)
    c, s = np.cos(a), np.sin(a)
    R = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1.0]])
    P = Q[rng.randint(0, m, n)] @ R.T + rng.uniform(-0.1, 0.1, 3) + 10.0 ** rng.uniform(-5, -2) * rng.standard_normal((n, 3))
    if rng.rand() < 0.3:
        P = P[P[:, 0] > np.median(P[:, 0]) - 0.2]          # partial overlap
    if rng.rand() < 0.3:
        k = len(P) // 20
        P[rng.permutation(len(P))[:k]] += rng.uniform(1, 3, (k, 3))   # outliers
    outs = []
    for search in ("grid", "f32", "auto") + (("exact",) if n * m < 1.5e10 else ()):
        e = icp_amd.Engine(search=search)
        e.set_target(Q); e.set_source(P); e.iterate(iters)
        outs.append((e.get_last()["idx"], e.get_source()))
        e.close()
    if not all(np.array_equal(outs[0][0], o[0]) and np.array_equal(outs[0][1], o[1]) for o in outs[1:]):
        bad += 1
        print("MISMATCH kind=%d n=%d m=%d iters=%d" % (kind, len(P), m, iters), flush=True)
    cases += 1
print("cases %d, mismatches %d" % (cases, bad), flush=True)
sys.exit(1 if bad else 0)
