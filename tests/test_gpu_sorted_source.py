"""options.sort_source: the engine stores a large source cloud in a spatial order of its own (include/icp_amd.h).  Clouds
and indices still go in and come out in the caller's row order; correspondences are the exact ones for the cloud at hand;
Pc / cov are summed over the stored order, so they agree with the unsorted engine to rounding, not to the bit; the stored
order is a function of the cloud alone (same bits from run to run, across search variants)."""
import numpy as np
import pytest

import icp_amd
from icp_amd import synth
from conftest import golden, relerr
from oracle import pyoracle as orc

pytestmark = pytest.mark.gpu


def run(P, Q, iters, search="auto", sort=True, hist=False):
    e = icp_amd.Engine(search=search, sort_source=sort)
    e.set_target(Q)
    e.set_source(P)
    out = []
    if hist:
        for _ in range(iters):
            e.iterate(1)
            out.append(e.get_last())
    else:
        e.iterate(iters)
        out.append(e.get_last())
    src = e.get_source()
    e.close()
    return out, src


def test_sorted_storage_keeps_the_callers_row_order_and_the_exact_indices(monkeypatch):
    """90k x 80k (the option's floor of 2^18 rows lowered for the test): every iteration's indices are the oracle's for the cloud that
    iteration saw (checked through the cloud the engine hands back), the first iteration's indices, Pc and cov equal the
    unsorted engine's up to the summation order, and the registered cloud comes back in the caller's row order.  At its real floor (300k rows) the option is
    checked once against the unsorted engine as well."""
    monkeypatch.setenv("ICP_AMD_SORT_MIN", "0")
    P, Q = synth.make_pair(90000, 80000, seed=77, angle=0.15)
    (a,), Pa = run(P, Q, 1, sort=True)
    (b,), Pb = run(P, Q, 1, sort=False)
    assert np.array_equal(a["idx"], b["idx"])                      # same cloud, same exact search, caller's row order
    assert np.allclose(a["Pc"], b["Pc"], rtol=0, atol=1e-15) and not np.array_equal(a["cov"], b["cov"])   # another summation order ...
    assert np.allclose(a["cov"], b["cov"], rtol=1e-12, atol=1e-12 * np.abs(b["cov"]).max())              # ... the same sums
    assert np.allclose(Pa, Pb, rtol=0, atol=1e-12)
    # the indices of a later iteration against the oracle on the cloud of that iteration
    e = icp_amd.Engine(sort_source=True)
    e.set_target(Q)
    e.set_source(P)
    e.iterate(6)
    cloud = e.get_source()            # the cloud the 7th search will see, in the caller's row order
    e.iterate(1)
    got = e.get_last()
    e.close()
    Qz = Q - orc.mean0(Q)
    want = orc.correspondences(cloud - got["Pc"], Qz, nthreads=8)
    assert np.array_equal(got["idx"], want)
    monkeypatch.delenv("ICP_AMD_SORT_MIN")
    P, Q = synth.make_pair(300000, 300000, seed=80)
    (a,), Pa = run(P, Q, 3, sort=True)
    (b,), Pb = run(P, Q, 3, sort=False)
    assert np.array_equal(a["idx"], b["idx"]) and not np.array_equal(a["cov"], b["cov"])
    assert np.allclose(Pa, Pb, rtol=0, atol=1e-12)


def test_sorted_storage_is_deterministic_and_the_same_for_every_search_variant(monkeypatch):
    monkeypatch.setenv("ICP_AMD_SORT_MIN", "0")
    P, Q = synth.make_pair(70000, 70000, seed=78)
    (a,), Pa = run(P, Q, 5, search="grid")
    (b,), Pb = run(P, Q, 5, search="grid")
    (c,), Pc_ = run(P, Q, 5, search="f32")
    for x, Px in ((b, Pb), (c, Pc_)):
        assert np.array_equal(a["idx"], x["idx"]) and np.array_equal(a["cov"], x["cov"]) and np.array_equal(a["R"], x["R"])
        assert np.array_equal(Pa, Px)


def test_sorted_storage_on_the_reference_fixtures(monkeypatch):
    """With the floor lowered (ICP_AMD_DEBUG sort_min=0) the option runs on the reference's own pairs: the indices of all 20
    iterations equal the reference's, Pc / cov / R / cloud within the usual bar -- the summation order moves nothing that
    matters."""
    monkeypatch.setenv("ICP_AMD_SORT_MIN", "0")
    monkeypatch.setenv("ICP_AMD_SORT_SOURCE", "1")
    for name in ("cow_tr1", "horse_tr1_x20", "bun045_000_x20"):
        g = golden(name)
        base = golden(str(g["inputs_from"])) if "inputs_from" in g else g
        P, Q, iters = (g["P"] if "P" in g else base["P"]), base["Q"], int(g["iters"])
        P_out, tr = icp_amd.icp_gpu_optimized(P, Q, iters, "-shared-loop", trace=True)
        assert int((tr["idx"] != g["idx"]).sum()) == 0, name
        assert relerr(tr["Pc"], g["Pc"]) < 1e-9 and relerr(tr["cov"], g["cov"]) < 1e-9 and relerr(tr["R"], g["R"]) < 1e-9
        assert relerr(P_out, g["P_out"]) < 1e-9
        monkeypatch.setenv("ICP_AMD_SORT_SOURCE", "0")
        P_plain, tr_plain = icp_amd.icp_gpu_optimized(P, Q, iters, "-shared-loop", trace=True)
        monkeypatch.setenv("ICP_AMD_SORT_SOURCE", "1")
        assert not np.array_equal(tr["cov"], tr_plain["cov"]), name      # the option did change the summation order


def test_sorted_storage_declines_where_it_cannot_or_need_not(monkeypatch):
    """Small clouds, strict order, a cloud whose rows all coincide (no order to speak of): the option is ignored and the
    results are the unsorted engine's, bit for bit."""
    P, Q = synth.make_pair(3000, 3000, seed=79)
    (a,), Pa = run(P, Q, 4, sort=True)
    (b,), Pb = run(P, Q, 4, sort=False)
    assert np.array_equal(a["cov"], b["cov"]) and np.array_equal(Pa, Pb)
    monkeypatch.setenv("ICP_AMD_SORT_MIN", "0")
    e1 = icp_amd.Engine(sort_source=True, strict_order=True)
    e0 = icp_amd.Engine(sort_source=False, strict_order=True)
    res = []
    for e in (e1, e0):
        e.set_target(Q)
        e.set_source(P)
        e.iterate(3)
        res.append((e.get_last(), e.get_source()))
        e.close()
    assert np.array_equal(res[0][0]["cov"], res[1][0]["cov"]) and np.array_equal(res[0][1], res[1][1])
    same = np.tile(np.array([[0.25, -0.5, 0.125]]), (9000, 1))
    (c,), Pc_ = run(same, Q, 2, sort=True)
    (d,), Pd = run(same, Q, 2, sort=False)
    assert np.array_equal(c["idx"], d["idx"]) and np.array_equal(Pc_, Pd)


def test_sorted_storage_follows_target_and_source_replacement(monkeypatch):
    """A target replaced under a sorted source (the row records are rebuilt from the stored order), then a source of another
    size (a new order): after each change the next search's indices are the oracle's for the cloud at hand, in the
    caller's row order -- also when AUTO answers by brute force (a target below the grid's floor)."""
    monkeypatch.setenv("ICP_AMD_SORT_MIN", "0")

    def check(e, Q):
        cloud = e.get_source()
        e.iterate(1)
        got = e.get_last()
        want = orc.correspondences(cloud - got["Pc"], Q - orc.mean0(Q), nthreads=8)
        assert np.array_equal(got["idx"], want)

    P1, Q1 = synth.make_pair(30000, 28000, seed=81)
    P2, Q2 = synth.make_pair(17000, 40000, seed=82, angle=0.3)
    _, Q3 = synth.make_pair(500, 900, seed=83)           # fewer than 1024 targets: AUTO searches by brute force
    e = icp_amd.Engine(sort_source=True)
    e.set_target(Q1)
    e.set_source(P1)
    e.iterate(3)
    check(e, Q1)
    e.set_target(Q2)
    check(e, Q2)
    e.iterate(2)
    check(e, Q2)
    e.set_source(P2)
    check(e, Q2)
    e.iterate(2)
    check(e, Q2)
    e.set_target(Q3)
    check(e, Q3)
    assert e.search_geometry()["variant"] != "grid"
    e.close()
