#!/usr/bin/env python3
"""Differential fuzz of the ORACLE (oracle/icp_oracle.c through oracle/pyoracle.py) against the unmodified reference
(oracle/_ref/ref_trace) -- CPU only, build container only.  The golden vectors pin the oracle on a dozen clouds; this pins it
on as many random adversarial ones as time allows: lattices with exact ties, duplicates, tight clusters, near-ties a few
ulp apart, wild scales and offsets, coplanar / collinear clouds, N != M.  Stage by stage from the reference's own
P_k: centroid, correspondence indices and covariance bit-exact; S, R, t to 1e-12 where the covariance is well conditioned.
    python tests/fuzz/fuzz_oracle.py [seconds] [seed]"""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import pyoracle as orc
from make_golden import TRACE, write_csv, read_trace


def relerr(a, b):
    s = max(np.abs(b).max(), 1e-300)
    return float(np.abs(np.asarray(a) - np.asarray(b)).max() / s)


def make_case(rng):
    n, m = int(rng.randint(1, 400)), int(rng.randint(1, 500))
    kind = rng.randint(8)
    if kind == 0:      # integer lattice, exact ties
        Q = rng.randint(-5, 6, (m, 3)).astype(np.float64)
        P = rng.randint(-5, 6, (n, 3)).astype(np.float64) + rng.choice([0.0, 0.5], (n, 3))
    elif kind == 1:    # duplicates in the target
        base = rng.uniform(-1, 1, (max(1, m // 3), 3))
        Q = base[rng.randint(0, len(base), m)]
        P = rng.uniform(-1, 1, (n, 3))
    elif kind == 2:    # tight clusters
        c = rng.uniform(-1, 1, (6, 3))
        Q = c[rng.randint(0, 6, m)] + rng.standard_normal((m, 3)) * 1e-9
        P = c[rng.randint(0, 6, n)] + rng.standard_normal((n, 3)) * 1e-3
    elif kind == 3:    # near-ties: targets in pairs a few ulp apart
        half = rng.uniform(-1, 1, ((m + 1) // 2, 3))
        Q = np.concatenate([half, np.nextafter(half, half + rng.choice([-1.0, 1.0], half.shape))])[:m]
        Q = Q[rng.permutation(len(Q))]
        P = rng.uniform(-1, 1, (n, 3))
    elif kind == 4:    # wild scale and offset
        s, o = 10.0 ** rng.uniform(-8, 8), 10.0 ** rng.uniform(-3, 5) * rng.choice([0.0, 1.0])
        Q = rng.uniform(-1, 1, (m, 3)) * s + o
        P = rng.uniform(-1, 1, (n, 3)) * s + o
    elif kind == 5:    # coplanar target
        Q = rng.uniform(-1, 1, (m, 3)); Q[:, 2] = 0.25
        P = rng.uniform(-1, 1, (n, 3))
    elif kind == 6:    # collinear source
        Q = rng.uniform(-1, 1, (m, 3))
        P = np.outer(rng.uniform(-1, 1, n), rng.uniform(-1, 1, 3)) + 0.1
    else:              # a rigidly moved, noisy copy (the usual registration input)
        Q = rng.uniform(-1, 1, (m, 3))
        a = rng.uniform(-0.5, 0.5)
        R = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])
        P = Q[rng.randint(0, m, n)] @ R.T + rng.uniform(-0.2, 0.2, 3) + rng.standard_normal((n, 3)) * 1e-3
    return np.ascontiguousarray(P), np.ascontiguousarray(Q), kind


def check(P, Q, iters, td):
    pf, qf, out = os.path.join(td, "P.csv"), os.path.join(td, "Q.csv"), os.path.join(td, "t.bin")
    write_csv(pf, P); write_csv(qf, Q)
    r = subprocess.run([TRACE, pf, qf, str(iters), out, "1"], capture_output=True)
    if r.returncode != 0:
        return "reference failed: " + r.stderr.decode()[-200:]
    g = read_trace(out)
    Qc = orc.mean0(Q)
    if not np.array_equal(Qc, g["Qc"]): return "Qc"
    Qz = orc.sub_row(Q, Qc)
    for it in range(iters):
        P_in = g["P_in"][it]
        if not np.all(np.isfinite(P_in)): return None      # the reference itself left the finite range (degenerate SVD): nothing to pin
        Pc = orc.mean0(P_in)
        if not np.array_equal(Pc, g["Pc"][it]): return "Pc at iteration %d" % it
        Pz = orc.sub_row(P_in, Pc)
        idx = orc.correspondences(Pz, Qz)
        if not np.array_equal(idx, g["idx"][it]): return "idx at iteration %d (%d differ)" % (it, int((idx != g["idx"][it]).sum()))
        cov = orc.cross_cov(Pz, Qz, idx)
        if not np.array_equal(cov, g["cov"][it]): return "cov at iteration %d" % it
        U, S, VT = orc.svd3(cov)
        Sr = g["S"][it]
        if np.all(np.isfinite(Sr)) and Sr[0] > 0:
            if np.abs(S - Sr).max() > 1e-12 * Sr[0]: return "S at iteration %d" % it
            if Sr[2] > 1e-6 * Sr[0]:
                R, t = orc.rt(U, VT, Pc, Qc)
                if relerr(R, g["R"][it]) > 1e-9: return "R at iteration %d: %g" % (it, relerr(R, g["R"][it]))
                scale = max(np.abs(g["t"][it]).max(), np.abs(Qc).max(), np.abs(Pc).max(), 1e-300)
                if np.abs(t - g["t"][it]).max() / scale > 1e-9: return "t at iteration %d" % it
        if np.all(np.isfinite(g["R"][it])) and np.all(np.isfinite(g["t"][it])):
            nxt = g["P_in"][it + 1] if it + 1 < iters else g["P_out"]
            if not np.array_equal(orc.apply(P_in, g["R"][it], g["t"][it]), nxt): return "apply at iteration %d" % it
    return None


def run(seconds, seed, verbose=True, max_cases=None):
    rng = np.random.RandomState(seed)
    t_end, t_say = time.time() + seconds, time.time() + 30
    cases, bad = 0, []
    with tempfile.TemporaryDirectory() as td:
        while time.time() < t_end and (max_cases is None or cases < max_cases):
            P, Q, kind = make_case(rng)
            why = check(P, Q, 3, td)
            cases += 1
            if why:
                bad.append((kind, P.shape, Q.shape, why))
                if verbose: print("MISMATCH kind %d  %s x %s: %s" % (kind, P.shape, Q.shape, why), flush=True)
            if verbose and time.time() > t_say:
                print("... %d cases, %d mismatches" % (cases, len(bad)), flush=True)
                t_say = time.time() + 30
    return cases, bad


if __name__ == "__main__":
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    cases, bad = run(seconds, seed)
    print("%d cases, %d mismatches" % (cases, len(bad)))
    sys.exit(1 if bad else 0)
