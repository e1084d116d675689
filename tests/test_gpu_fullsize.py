"""GPU: the BASELINE configs at their FULL sizes and iteration counts against vectors taken from the reference itself
(tests/golden/make_golden.py: oracle/_ref/ref_trace = the reference's unmodified CPU sources).

 * configs[1]: horse 48 485 x 48 485 and bunny 40 097 x 40 256 (N != M), 20 iterations of the reference loop each.
 * configs[2]: synthetic 100k x 100k, 30 iterations; configs[3]: 1M x 1M.  The reference cannot walk these loops (minutes
   to hours per iteration), so its get_correspondence_indices (src/cpu/icp.cpp:72-93) answered for sampled rows against
   ALL targets at iterations 1 / 15 / 30 (100k) and 1 / 3 (1M), on the running cloud of the oracle port.  Two checks:
   kernel level -- the stage-level calls on exactly the fixture's rows, un-seeded and SEEDED (the kernel instantiation the
   benchmark times), must reproduce the reference's indices bit for bit, unconditionally; loop level -- the engine's own
   iterations must agree on those rows (its running cloud differs from the oracle port's in the last bits of the reduction
   sums, which could only flip a correspondence between two targets whose keys are within an ulp; none does on these rows).
"""
import numpy as np
import pytest

import icp_amd
from conftest import golden, relerr, relerr_t, stored_clouds
from icp_amd import synth
from oracle import pyoracle as orc

pytestmark = pytest.mark.gpu
TOL = 1e-9


@pytest.mark.parametrize("name", ["horse_tr1_x20", "bun045_000_x20", "horse_tr2_x20"])
def test_config2_sample_clouds_20_iterations(name):
    g = golden(name)
    base = golden(str(g["inputs_from"]))
    P, Q, iters = (g["P"] if "P" in g else base["P"]), base["Q"], int(g["iters"])   # horse_tr2: its own source, horse_tr1's target
    assert iters == 20
    P_out, tr = icp_amd.icp_gpu_optimized(P, Q, iters, "-shared-loop", trace=True)
    mism = int((tr["idx"] != g["idx"]).sum())
    assert mism == 0, "%s: %d correspondence mismatches over %d iterations" % (name, mism, iters)
    assert relerr(tr["Pc"], g["Pc"]) < TOL and relerr(tr["cov"], g["cov"]) < TOL and relerr(tr["R"], g["R"]) < TOL
    for it in range(iters):
        assert relerr_t(tr["t"][it], g["t"][it], g["Qc"], g["Pc"][it]) < TOL
    assert relerr(P_out, g["P_out"]) < TOL
    # the same 20 iterations through the brute-force search of the headline benchmark: bit-identical to the default path
    e = icp_amd.Engine(search="f32")
    e.set_target(Q)
    e.set_source(P)
    e.iterate(iters)
    assert np.array_equal(e.get_last()["idx"], g["idx"][-1]) and np.array_equal(e.get_source(), P_out)
    e.close()
    # stage level on the running cloud the reference itself held at iterations 2, 11 and 20
    Qz = orc.sub_row(Q, g["Qc"])
    for it, P_in in stored_clouds(g, P.shape):
        for search in ("f32", "grid"):
            idx = icp_amd.correspondences(P_in, g["Pc"][it], Qz, search=search, q_splits=2)
            assert np.array_equal(idx, g["idx"][it]), (name, it, search)
            seeded = icp_amd.correspondences_seeded(P_in, g["Pc"][it], Qz, g["idx"][it - 1], search=search, q_splits=3)
            assert np.array_equal(seeded, g["idx"][it]), (name, it, search, "seeded")


@pytest.mark.parametrize("name", ["bun180_000", "bun270_000", "bun315_000"])
def test_config2_remaining_bunny_scans_one_iteration(name):
    """The rest of data/icp: bun180 / bun270 / bun315 against bun000 (N != M: 40 251 / 31 701 / 35 336 source rows, 40 256
    targets), one iteration of the reference loop each -- every search variant at stage level, then the engine (AUTO = grid,
    and the brute-force search of the benchmark) for that iteration."""
    g = golden(name)
    P, Q, Qc = g["P"], golden(str(g["inputs_from"]))["Q"], g["Qc"]
    assert len(P) != len(Q)
    want = g["idx"][0]
    Qz = orc.sub_row(Q, Qc)
    assert relerr(icp_amd.centroid(Q), Qc) < 1e-13 and relerr(icp_amd.centroid(P), g["Pc"][0]) < 1e-13
    for search, splits in (("f32", 1), ("f32", 7), ("grid", 1), ("fma", 3), ("exact", 2)):
        idx = icp_amd.correspondences(P, g["Pc"][0], Qz, search=search, q_splits=splits)
        assert np.array_equal(idx, want), (name, search, splits, int((idx != want).sum()))
    assert relerr(icp_amd.cross_cov(P, g["Pc"][0], Qz, want), g["cov"][0]) < 1e-12
    for search in ("auto", "f32"):
        e = icp_amd.Engine(search=search)
        e.set_target(Q)
        e.set_source(P)
        e.iterate(1)
        last = e.get_last()
        e.close()
        assert np.array_equal(last["idx"], want), (name, search)
        assert relerr(last["Pc"], g["Pc"][0]) < TOL and relerr(last["cov"], g["cov"][0]) < TOL
        assert relerr(last["R"], g["R"][0]) < TOL and relerr(last["S"], g["S"][0]) < TOL
        assert relerr_t(last["t"], g["t"][0], Qc, g["Pc"][0]) < TOL


def _rows_fixture(name):
    g = golden(name)
    n = int(g["n"])
    P, Q = synth.make_pair(n, n)
    return g, P, Q, g["rows"], [int(k) for k in g["iters_at"]]


@pytest.mark.parametrize("name,groups", [("syn100k_rows", "2"), ("syn100k_rows", "4"), ("syn1m_rows", "4")])
def test_full_size_rows_kernel_level_against_the_reference(name, groups, monkeypatch):
    """Un-seeded and seeded stage-level launches on the fixture's own rows: bit-exact, every variant.  ICP_AMD_MFMA_G
    selects the source groups per wave of the matrix-pipe filter (4 = what the 1M x 1M benchmark launches)."""
    monkeypatch.setenv("ICP_AMD_MFMA_G", groups)
    g, P, Q, rows, iters_at = _rows_fixture(name)
    Qz = orc.sub_row(Q, g["Qc"])
    assert np.array_equal(g["Qc"], orc.mean0(Q))
    for k in iters_at:
        P_rows, Pc, want = g["P_rows_%d" % k], g["Pc_%d" % k], g["idx_%d" % k]
        if k == 0:
            assert np.array_equal(P_rows, P[rows])
        for search, splits in (("f32", 1), ("f32", 9), ("grid", 1), ("fma", 4), ("exact", 2)):
            if len(Q) > 200000 and search == "exact" and k > 0:
                continue
            if k == 0:
                got = icp_amd.correspondences(P_rows, Pc, Qz, search=search, q_splits=splits)   # brute force: seeded by the grid
                if search == "f32":
                    monkeypatch.setenv("ICP_AMD_GRID_SEED", "0")                                 # ... and truly un-seeded
                    raw = icp_amd.correspondences(P_rows, Pc, Qz, search=search, q_splits=splits)
                    monkeypatch.delenv("ICP_AMD_GRID_SEED")
                    assert np.array_equal(raw, want), (name, k, search, splits, "un-seeded")
            else:
                got = icp_amd.correspondences_seeded(P_rows, Pc, Qz, g["seed_%d" % k], search=search, q_splits=splits)
            assert np.array_equal(got, want), (name, k, search, splits, int((got != want).sum()))
        if k > 0:   # a seed that is wrong for every row must not change the answer either
            bad = np.roll(g["seed_%d" % k], 7)
            assert np.array_equal(icp_amd.correspondences_seeded(P_rows, Pc, Qz, bad, search="f32", q_splits=5), want)


@pytest.mark.parametrize("name,search", [("syn100k_rows", "auto"), ("syn100k_rows", "f32"), ("syn100k_rows", "grid"),
                                         ("syn1m_rows", "f32"), ("syn1m_rows", "auto"), ("syn1m_rows", "auto+sort"),
                                         ("syn1m_rows", "f32+sort")])
def test_full_size_loop_agrees_with_the_reference_on_sampled_rows(name, search):
    """configs[2] for its 30 iterations (checked at 1, 15, 30) and configs[3] (1 and 3: un-seeded and seeded launches of the
    kernel the benchmark times) through the engine; configs[3] also with the source stored in spatial order
    (icp_amd_options.sort_source: the rows and their indices still come back in the caller's order)."""
    g, P, Q, rows, iters_at = _rows_fixture(name)
    sort = search.endswith("+sort")
    search = search.split("+")[0]
    e = icp_amd.Engine(search=search, sort_source=sort)
    e.set_target(Q)
    e.set_source(P)
    assert relerr(e.target_centroid(), g["Qc"]) < 1e-13
    done = 0
    for k in iters_at:
        e.iterate(k + 1 - done)
        done = k + 1
        last = e.get_last()
        assert relerr(last["Pc"], g["Pc_%d" % k]) < 1e-12
        mism = int((last["idx"][rows] != g["idx_%d" % k]).sum())
        assert mism == 0, (name, search, k, mism)
    geo = e.search_geometry()
    assert geo["variant"] == ("grid" if search in ("auto", "grid") else "f32")
    e.close()
