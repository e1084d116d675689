"""GPU tests of the optional hooks: distance rejection in the cross-covariance and the
convergence-controlled loop.  Off by default -- then nothing changes; on, they are checked against the same loop
composed from the oracle's stages (the reference leaves these hooks inert: src/cpu/icp.cpp:108-124, :213)."""
import numpy as np
import pytest

import icp_amd
from conftest import relerr, relerr_t
from icp_amd import synth
from oracle import pyoracle as orc

pytestmark = pytest.mark.gpu
TOL = 1e-9


def oracle_loop_with_rejection(P, Q, iters, dmax):
    """icp() of the oracle with weight = [d2(p - q) <= dmax^2] in compute_cross_variance (d2 un-fused, as everywhere)."""
    Qc = orc.mean0(Q)
    Qz = orc.sub_row(Q, Qc)
    P = P.copy()
    out = []
    for _ in range(iters):
        Pc = orc.mean0(P)
        Pz = orc.sub_row(P, Pc)
        idx = orc.correspondences(Pz, Qz, nthreads=4)
        diff = Pz - Qz[idx]
        d2 = (diff[:, 0] * diff[:, 0] + diff[:, 1] * diff[:, 1]) + diff[:, 2] * diff[:, 2]
        keep = d2 <= dmax * dmax
        cov = orc.cross_cov(np.ascontiguousarray(Pz[keep]), Qz, np.ascontiguousarray(idx[keep]))
        U, S, VT = orc.svd3(cov)
        R, t = orc.rt(U, VT, Pc, Qc)
        out.append(dict(idx=idx, cov=cov, R=R, t=t, Pc=Pc, used=int(keep.sum()), rms=float(np.sqrt(d2[keep].sum() / max(1, keep.sum())))))
        P = orc.apply(P, R, t)
    return P, out, Qc


def test_distance_rejection_matches_the_hooked_oracle_loop():
    P, Q = synth.make_pair(4000, 5000, seed=21)
    rng = np.random.RandomState(2)
    P[::9] += rng.uniform(0.3, 0.8, (len(P[::9]), 3))            # a ninth of the points are gross outliers
    dmax = 0.12
    want_P, want, Qc = oracle_loop_with_rejection(P, Q, 4, dmax)
    for search in ("auto", "exact"):
        e = icp_amd.Engine(search=search)
        e.set_target(Q)
        e.set_source(P)
        e.set_reject_distance(dmax)
        for it in range(4):
            e.iterate(1)
            last, mo = e.get_last(), e.last_motion()
            w = want[it]
            assert np.array_equal(last["idx"], w["idx"]), (search, it)
            assert mo["pairs_used"] == w["used"] and 0 < w["used"] < len(P)
            assert relerr(last["cov"], w["cov"]) < TOL and relerr(last["R"], w["R"]) < TOL
            assert relerr_t(last["t"], w["t"], Qc, w["Pc"]) < TOL
            assert abs(last["rms"] - w["rms"]) < 1e-12 * max(1.0, w["rms"])
        assert relerr(e.get_source(), want_P) < TOL
        e.close()


def test_hooks_off_change_nothing_and_can_be_switched_off_again():
    P, Q = synth.make_pair(6000, 6000, seed=8)
    ref = icp_amd.icp_gpu_optimized(P, Q, 3)
    e = icp_amd.Engine()
    e.set_target(Q)
    e.set_source(P)
    e.set_reject_distance(0.05)
    e.set_reject_distance(0.0)                                     # <= 0: off
    e.iterate(3)
    assert e.last_motion()["pairs_used"] == len(P)
    assert np.array_equal(e.get_source(), ref)
    e.set_source(P)
    e.set_reject_distance(float("inf"))
    e.iterate(3)
    assert np.array_equal(e.get_source(), ref)
    e.close()


def test_iterate_until_stops_at_convergence_and_equals_a_fixed_count():
    P, Q = synth.make_pair(20000, 20000, seed=3)
    e = icp_amd.Engine()
    e.set_target(Q)
    e.set_source(P)
    done = e.iterate_until(200, 1e-10, 1e-10, check_every=1)
    mo = e.last_motion()
    P_until = e.get_source()
    assert 2 <= done < 200, done
    assert mo["delta_rot"] <= 1e-10 and mo["delta_trans"] <= 1e-10
    e.set_source(P)
    e.iterate(done - 1)
    prev = e.last_motion()
    assert prev["delta_rot"] > 1e-10 or prev["delta_trans"] > 1e-10   # it did not stop late
    e.iterate(1)
    assert np.array_equal(e.get_source(), P_until)                     # the test changes no result
    e.set_source(P)
    done4 = e.iterate_until(200, 1e-10, 1e-10, check_every=4)
    assert done4 % 4 == 0 and done <= done4 < done + 4
    assert e.iterate_until(3, 0.0, 0.0) == 3                           # tolerances 0: runs to the cap unless exactly fixed
    e.close()
