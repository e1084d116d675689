#!/usr/bin/env python3
"""Fuzz of the optional hooks: random rejection distances against the loop composed from the oracle's stages, and
iterate_until against fixed-count iteration."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import icp_amd
from icp_amd import synth
from test_gpu_hooks import oracle_loop_with_rejection
from conftest import relerr, relerr_t

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t_end = time.time() + budget
t_say = time.time() + 30
cases = bad = 0
while time.time() < t_end:
    if time.time() > t_say:
        print("... %d cases, %d mismatches so far" % (cases, bad), flush=True)
        t_say = time.time() + 30
    n, m, iters = int(rng.randint(50, 6000)), int(rng.randint(50, 6000)), int(rng.randint(1, 5))
    P, Q = synth.make_pair(n, max(n, m), seed=int(rng.randint(1 << 30)))
    Q = Q[:m] if m < len(Q) else Q
    k = max(1, n // int(rng.randint(3, 20)))
    P[rng.permutation(n)[:k]] += rng.uniform(0.2, 1.0, (k, 3))
    dmax = float(10.0 ** rng.uniform(-2.5, 0.3))
    want_P, want, Qc = oracle_loop_with_rejection(P, Q, iters, dmax)
    e = icp_amd.Engine(search=str(rng.choice(["auto", "f32", "exact"])))
    e.set_target(Q); e.set_source(P); e.set_reject_distance(dmax)
    ok = True
    why = []
    for it in range(iters):
        e.iterate(1)
        last, mo = e.get_last(), e.last_motion()
        w = want[it]
        if w["used"] < 50:   # a handful of pairs: the covariance is (nearly) rank deficient, R is not determined -- stop comparing
            break
        if not np.array_equal(last["idx"], w["idx"]): why.append("idx it%d (%d rows)" % (it, int((last["idx"] != w["idx"]).sum())))
        if mo["pairs_used"] != w["used"]: why.append("used it%d %d vs %d" % (it, mo["pairs_used"], w["used"]))
        if w["used"] >= 10 and relerr(last["cov"], w["cov"]) >= 1e-9: why.append("cov it%d %.2e" % (it, relerr(last["cov"], w["cov"])))
        if w["used"] >= 10 and relerr(last["R"], w["R"]) >= 1e-9: why.append("R it%d %.2e S=%s" % (it, relerr(last["R"], w["R"]), last["S"]))
    if all(w["used"] >= 50 for w in want) and relerr(e.get_source(), want_P) >= 1e-8:
        why.append("P %.2e" % relerr(e.get_source(), want_P))
    ok = not why
    e.set_reject_distance(0.0)
    e.set_source(P)
    done = e.iterate_until(30, 1e-9, 1e-9, check_every=int(rng.randint(1, 4)))
    P_until = e.get_source()
    e.set_source(P); e.iterate(done)
    if not np.array_equal(e.get_source(), P_until): why.append("iterate_until != iterate(%d)" % done)
    ok = not why
    e.close()
    if not ok:
        bad += 1
        print("MISMATCH n=%d m=%d iters=%d dmax=%g: %s" % (n, m, iters, dmax, "; ".join(why[:4])), flush=True)
    cases += 1
print("cases %d, mismatches %d" % (cases, bad), flush=True)
sys.exit(1 if bad else 0)
