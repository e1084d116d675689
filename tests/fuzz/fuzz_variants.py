#!/usr/bin/env python3
"""Differential fuzzing of the search variants (exact / fma / f32 / grid, several split counts) and the oracle on small
adversarial clouds: lattices, duplicates, clusters, near-ties a few ulp apart, wild scales.  Exit code 1 on a mismatch."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import icp_amd
from oracle import pyoracle as orc

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t_end = time.time() + budget
cases = bad = 0
t_say = time.time() + 30
while time.time() < t_end:
    if time.time() > t_say:
        print("... %d cases, %d mismatches so far" % (cases, bad), flush=True)
        t_say = time.time() + 30
    n, m = int(rng.randint(1, 3000)), int(rng.randint(1, 5000))
    kind = rng.randint(6)
    if kind == 0:      # integer lattice, exact ties
        Q = rng.randint(-6, 7, (m, 3)).astype(np.float64)
        P = rng.randint(-6, 7, (n, 3)).astype(np.float64) + rng.choice([0.0, 0.5], (n, 3))
    elif kind == 1:    # duplicates in the target
        base = rng.uniform(-1, 1, (max(1, m // 3), 3))
        Q = base[rng.randint(0, len(base), m)]
        P = rng.uniform(-1, 1, (n, 3))
    elif kind == 2:    # tight clusters
        c = rng.uniform(-1, 1, (8, 3))
        Q = c[rng.randint(0, 8, m)] + 1e-6 * rng.standard_normal((m, 3))
        P = c[rng.randint(0, 8, n)] + 1e-3 * rng.standard_normal((n, 3))
    elif kind == 3:    # near-ties: targets mirrored about the source points, perturbed by a few ulp
        P = rng.uniform(-1, 1, (n, 3))
        k = min(n, m // 2)
        off = rng.uniform(0.01, 0.1, (k, 3))
        Q = np.concatenate([P[:k] + off, P[:k] - off * (1 + rng.randint(-4, 5, (k, 1)) * 2.0 ** -52), rng.uniform(-1, 1, (m - 2 * k, 3))])
        Q = Q[rng.permutation(len(Q))][:m] if len(Q) >= m else Q
        m = len(Q)
    elif kind == 4:    # wild scale / offset
        s, o = 10.0 ** rng.uniform(-30, 30), rng.choice([0.0, 1.0, 1e4]) 
        Q = rng.uniform(-1, 1, (m, 3)) * s + o * s
        P = rng.uniform(-1, 1, (n, 3)) * s + o * s
    else:              # flat / thin clouds
        Q = rng.uniform(-1, 1, (m, 3)) * np.array([1.0, rng.choice([0.0, 1e-9, 1.0]), rng.choice([0.0, 1.0])])
        P = rng.uniform(-1, 1, (n, 3)) * np.array([1.0, 1.0, rng.choice([0.0, 1.0])])
    Pc = orc.mean0(P)
    Qz = Q - orc.mean0(Q)
    if rng.randint(12) == 0:   # a few non-finite coordinates in single rows (behind the centring, so that the other rows
        P, Q = P.copy(), Q.copy()   # stay finite): every variant must do with them what the oracle does
        for _ in range(int(rng.randint(1, 4))):
            X = P if rng.randint(2) else Qz
            X[rng.randint(len(X)), rng.randint(3)] = rng.choice([np.nan, np.inf, -np.inf])
    want = orc.correspondences(P - Pc, Qz, nthreads=8)
    # half the cases run the brute-force launches truly un-seeded (the cooperative exact path of the matrix filter), the
    # other half seeded from the grid's answer like the engine's first iteration
    os.environ["ICP_AMD_GRID_SEED"] = "0" if rng.rand() < 0.5 else "1"
    # the grid's processing order of the rows (spatial / storage, and where it switches) and its cell size never matter
    os.environ["ICP_AMD_GRID_ORDER_SWITCH"] = str(rng.choice([0.0, 0.125, 4.0, 1e30]))
    os.environ["ICP_AMD_GRID_PER_CELL"] = str(rng.choice([0.5, 1.0, 2.0, 2.0, 6.0]))
    for search, splits in (("exact", 1), ("fma", int(rng.randint(1, 9))), ("f32", int(rng.randint(1, 9))), ("grid", 1)):
        got = icp_amd.correspondences(P, Pc, Qz, search=search, q_splits=splits)
        if not np.array_equal(got, want):
            bad += 1
            print("MISMATCH kind=%d n=%d m=%d search=%s splits=%d: %d rows" % (kind, n, m, search, splits, int((got != want).sum())), flush=True)
    # seeded iterations through the engine: all variants against exact
    outs = []
    n_it = int(rng.randint(2, 9))
    for search in ("exact", "f32", "grid"):
        e = icp_amd.Engine(search=search)
        e.set_target(Q); e.set_source(P); e.iterate(n_it)
        outs.append(e.get_last()["idx"]); e.close()
    if not (np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])):
        bad += 1
        print("ENGINE MISMATCH kind=%d n=%d m=%d" % (kind, n, m), flush=True)
    cases += 1
print("cases %d, mismatches %d" % (cases, bad), flush=True)
sys.exit(1 if bad else 0)
