"""GPU parity tests of the grid search (icp_amd/csrc/grid.hip): it must return exactly the
correspondences of the brute-force scan -- the reference's golden indices, the oracle's, and the other variants' --
on every fixture and on the shapes that stress its stop rule (ties, flat clouds, far-apart clouds, huge and tiny
scales, non-finite coordinates)."""
import numpy as np
import pytest

import icp_amd
from conftest import ICP_CASES, golden, relerr
from icp_amd import synth
from oracle import pyoracle as orc

pytestmark = pytest.mark.gpu
TOL = 1e-9


@pytest.mark.parametrize("name", ICP_CASES)
def test_grid_stagewise_against_reference(name):
    g = golden(name)
    Qz = g["Q"] - g["Qc"]
    for slot, it in enumerate(g["P_in_iters"]):
        idx = icp_amd.correspondences(g["P_in"][slot], g["Pc"][it], Qz, search="grid")
        assert np.array_equal(idx, g["idx"][it]), (name, it, int((idx != g["idx"][it]).sum()))


@pytest.mark.parametrize("name", ICP_CASES)
def test_grid_whole_loop_against_reference(name):
    """Seeded iterations included: the engine runs every iteration on the grid."""
    g = golden(name)
    iters = int(g["iters"])
    e = icp_amd.Engine(search="grid")
    e.set_target(g["Q"])
    e.set_source(g["P"])
    assert e.search_geometry()["variant"] == "grid"
    for it in range(iters):
        e.iterate(1)
        last = e.get_last()
        assert np.array_equal(last["idx"], g["idx"][it]), (name, it)
        assert relerr(last["cov"], g["cov"][it]) < TOL
    P_out = e.get_source()
    e.close()
    assert relerr(P_out, g["P_out"]) < TOL


def test_grid_ties_and_lattice():
    g = golden("ties_corr")
    idx = icp_amd.correspondences(g["Pz"], np.zeros(3), g["Qz"], search="grid")
    assert np.array_equal(idx, g["idx"])
    # integer lattice: masses of exactly equal distances, the lowest index must win whatever the visiting order
    ax = np.arange(-6, 7, dtype=np.float64)
    Q = np.stack(np.meshgrid(ax, ax, ax, indexing="ij"), -1).reshape(-1, 3)
    Q = Q[np.random.RandomState(2).permutation(len(Q))]
    P = np.concatenate([Q[:500] + 0.5, Q[500:900] + np.array([0.5, 0.0, 0.0]), Q[900:1200]])
    Qz = Q - orc.mean0(Q)
    Pc = orc.mean0(P)
    want = orc.correspondences(P - Pc, Qz)
    for search in ("grid", "exact"):
        assert np.array_equal(icp_amd.correspondences(P, Pc, Qz, search=search), want), search


def test_grid_edge_shapes_and_placements():
    rng = np.random.RandomState(3)
    for n, m in ((1, 1), (1, 5), (7, 1), (63, 65), (1025, 1023), (300, 2049), (2050, 257)):
        P, Q = rng.randn(n, 3), rng.randn(m, 3)
        Pc, Qz = orc.mean0(P), Q - orc.mean0(Q)
        assert np.array_equal(icp_amd.correspondences(P, Pc, Qz, search="grid"), orc.correspondences(P - Pc, Qz)), (n, m)
    Q = rng.uniform(-1, 1, (4000, 3))
    Qz = Q - orc.mean0(Q)
    cases = {
        "far": rng.uniform(-1, 1, (500, 3)) + np.array([40.0, -25.0, 3.0]),     # every query outside the bounding box
        "wide": rng.uniform(-30, 30, (500, 3)),                                  # inside and far outside
        "on_targets": Q[rng.permutation(4000)[:500]],                            # d = 0 for every query
        "tiny_offsets": Q[:500] + rng.uniform(-1e-14, 1e-14, (500, 3)),
    }
    for name, P in cases.items():
        Pc = np.zeros(3)
        assert np.array_equal(icp_amd.correspondences(P, Pc, Qz, search="grid"), orc.correspondences(P, Qz)), name
    # clouds that are flat along one or two axes, or a single repeated point
    for flat in ((2,), (0, 2), (0, 1, 2)):
        Qf = Q.copy()
        Qf[:, list(flat)] = 0.25
        Qfz = Qf - orc.mean0(Qf)
        P = rng.uniform(-1.5, 1.5, (700, 3))
        assert np.array_equal(icp_amd.correspondences(P, np.zeros(3), Qfz, search="grid"), orc.correspondences(P, Qfz)), flat


def test_grid_scales_and_nonfinite():
    P, Q = synth.make_pair(5000, 6000, seed=77)
    for scale, shift in ((1.0, 0.0), (1e-6, 0.0), (1e6, 0.0), (1.0, 1e4), (1e-3, 50.0), (1e20, 0.0), (1e-20, 0.0), (1e150, 0.0),
                         (1e-150, 0.0)):
        Ps, Qs = P * scale + shift, Q * scale + shift
        outs = []
        for search in ("exact", "grid"):
            e = icp_amd.Engine(search=search)
            e.set_target(Qs)
            e.set_source(Ps)
            e.iterate(3)
            outs.append((e.get_last()["idx"], e.get_source()))
            e.close()
        assert np.array_equal(outs[0][0], outs[1][0]), (scale, shift)
        assert np.array_equal(outs[0][1], outs[1][1], equal_nan=True), (scale, shift)   # 1e150: cov overflows in both
    # non-finite source rows have no correspondence in either variant; a non-finite target disables the grid
    Pz, Qz = P - orc.mean0(P), Q - orc.mean0(Q)
    Pn = Pz.copy()
    Pn[5, 0], Pn[17, 1], Pn[99, 2] = np.nan, np.inf, -np.inf
    a = icp_amd.correspondences(Pn, np.zeros(3), Qz, search="exact")
    b = icp_amd.correspondences(Pn, np.zeros(3), Qz, search="grid")
    assert np.array_equal(a, b) and a[5] == -1 and a[17] == -1 and a[99] == -1
    Qn = Qz.copy()
    Qn[7, 1], Qn[9, 0] = np.inf, np.nan
    a = icp_amd.correspondences(Pz, np.zeros(3), Qn, search="exact")
    b = icp_amd.correspondences(Pz, np.zeros(3), Qn, search="grid")
    assert np.array_equal(a, b)


def test_grid_full_sizes_match_brute_force():
    """BASELINE configs 3 and 4 (100k x 100k, 1M x 1M): three iterations on the grid equal three brute-force ones,
    bit for bit (indices, covariance, cloud)."""
    for n in (100000, 1000000):
        P, Q = synth.make_pair(n, n)
        outs = {}
        for search in ("f32", "grid"):
            e = icp_amd.Engine(search=search)
            e.set_target(Q)
            e.set_source(P)
            e.iterate(3)
            outs[search] = (e.get_last(), e.get_source())
            e.close()
        a, b = outs["f32"], outs["grid"]
        assert np.array_equal(a[0]["idx"], b[0]["idx"]), n
        assert np.array_equal(a[0]["cov"], b[0]["cov"]) and np.array_equal(a[1], b[1]), n


def test_grid_overflow_queue_and_auto_fallback(monkeypatch):
    """Points whose box is too large for one lane are finished by the overflow kernel (one workgroup each); with tiny
    caps every point takes that route.  Partial overlap / outliers exercise it with the default caps.  AUTO falls back
    to brute force while most points overflow.  Results never change."""
    rng = np.random.RandomState(11)
    P, Q = synth.make_pair(6000, 7000, seed=5)
    Pc, Qz = orc.mean0(P), Q - orc.mean0(Q)
    want = orc.correspondences(P - Pc, Qz, nthreads=4)
    monkeypatch.setenv("ICP_AMD_GRID_DEFER", "always")
    for rowcap, evalcap in ((1, 8), (64, 1), (2, 100000)):
        monkeypatch.setenv("ICP_AMD_GRID_ROWCAP", str(rowcap))
        monkeypatch.setenv("ICP_AMD_GRID_EVALCAP", str(evalcap))
        assert np.array_equal(icp_amd.correspondences(P, Pc, Qz, search="grid"), want), (rowcap, evalcap)
        e = icp_amd.Engine(search="grid")
        e.set_target(Q)
        e.set_source(P)
        e.iterate(3)
        got = e.get_last()["idx"]
        e.close()
        ref = orc.icp(P, Q, 3, trace=True, nthreads=4)
        assert np.array_equal(got, ref["idx"][-1]), (rowcap, evalcap)
    monkeypatch.delenv("ICP_AMD_GRID_DEFER")
    for rowcap, evalcap in ((1, 8), (2, 16)):       # default rule: boxes up to 16x the caps are walked when the whole wave has them
        monkeypatch.setenv("ICP_AMD_GRID_ROWCAP", str(rowcap))
        monkeypatch.setenv("ICP_AMD_GRID_EVALCAP", str(evalcap))
        assert np.array_equal(icp_amd.correspondences(P, Pc, Qz, search="grid"), want), (rowcap, evalcap)
    monkeypatch.delenv("ICP_AMD_GRID_ROWCAP")
    monkeypatch.delenv("ICP_AMD_GRID_EVALCAP")
    # outliers: a tenth of the source points far away from every target, and a source that overlaps half the target
    Po = P.copy()
    Po[::10] += rng.uniform(3, 6, (600, 3))
    Ph = P[P[:, 0] > np.median(P[:, 0])]
    for name, Px in (("outliers", Po), ("half", Ph)):
        outs = []
        for search in ("exact", "grid", "auto"):
            e = icp_amd.Engine(search=search)
            e.set_target(Q)
            e.set_source(Px)
            e.iterate(4)
            outs.append((e.get_last()["idx"], e.get_source()))
            e.close()
        for o in outs[1:]:
            assert np.array_equal(outs[0][0], o[0]) and np.array_equal(outs[0][1], o[1]), name
    # clouds at right angles (a long thin target, the source turned by 90 degrees): most boxes are huge, AUTO must
    # still return the brute-force result while it switches between the grid and the tiled kernel
    Ql = rng.uniform(-1, 1, (20000, 3)) * np.array([20.0, 1.0, 1.0])
    Pl = Ql[rng.permutation(20000)][:, [1, 0, 2]] + 0.01 * rng.randn(20000, 3)
    outs, stats = [], {}
    for search in ("f32", "auto", "grid"):
        e = icp_amd.Engine(search=search)
        e.set_target(Ql)
        e.set_source(Pl)
        for _ in range(10):   # (the queue passes n / 8, where AUTO switches, around the fifth search; how many rows a wave hands over
            e.iterate(1)      # depends on which rows share it, i.e. on the order the walking rows were packed in)
            e.sync()
        outs.append((e.get_last()["idx"], e.get_source()))
        stats[search] = e.grid_stats()
        e.close()
    for o in outs[1:]:
        assert np.array_equal(outs[0][0], o[0]) and np.array_equal(outs[0][1], o[1])
    assert stats["f32"]["grid_launches"] == 0 and stats["grid"]["brute_force_launches"] == 0
    assert stats["auto"]["grid_launches"] >= 1 and stats["auto"]["brute_force_launches"] >= 1, stats
    # the same at a scale where the fp32 filter's bound means nothing: behind the grid the brute-force kernels learn the
    # target's range only at their first launch, and must then pick the fp64 filter
    outs = []
    for search in ("exact", "auto"):
        e = icp_amd.Engine(search=search)
        e.set_target(Ql * 1e20)
        e.set_source(Pl * 1e20)
        for _ in range(6):   # the fallback decision reads the queue length of a FINISHED search: let each one finish
            e.iterate(1)
            e.sync()
        outs.append((e.get_last()["idx"], e.get_source(), e.grid_stats()))
        e.close()
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    assert outs[1][2]["brute_force_launches"] >= 1, outs[1][2]


def test_grid_state_follows_target_and_source_replacement():
    """AUTO decides per target: a small target (brute force) replaced by a large one under the same source switches to
    the grid, and back; a new source under the same target rebuilds the per-source records.  Results = brute force."""
    P, Qs = synth.make_pair(5000, 900, seed=41)
    _, Ql = synth.make_pair(5000, 9000, seed=42)
    e = icp_amd.Engine()
    e.set_target(Qs)
    e.set_source(P)
    assert e.search_geometry()["variant"] == "f32"
    e.iterate(2)
    e.set_target(Ql)
    assert e.search_geometry()["variant"] == "grid"
    P_mid = e.get_source()
    e.iterate(3)
    got = (e.get_last()["idx"], e.get_source())
    assert e.grid_stats()["grid_launches"] == 3
    e.set_target(Qs)
    assert e.search_geometry()["variant"] == "f32"
    e.set_target(Ql)
    e.set_source(P_mid)
    e.iterate(3)
    again = (e.get_last()["idx"], e.get_source())
    e.close()
    b = icp_amd.Engine(search="exact")
    b.set_target(Ql)
    b.set_source(P_mid)
    b.iterate(3)
    want = (b.get_last()["idx"], b.get_source())
    b.close()
    for r in (got, again):
        assert np.array_equal(r[0], want[0]) and np.array_equal(r[1], want[1])


def test_grid_on_surfaces_refines_its_cells_and_keeps_the_results(monkeypatch):
    """Scan-like clouds (points on a surface): the cell edge is refined from the occupied-cell count; with and without
    the refinement the correspondences equal the brute-force ones."""
    rng = np.random.RandomState(12)
    v = rng.standard_normal((30000, 3))
    Q = v / np.linalg.norm(v, axis=1)[:, None] * np.array([1.0, 0.7, 0.5])
    c, s = np.cos(0.04), np.sin(0.04)
    P = Q[rng.permutation(30000)[:20000]] @ np.array([[c, -s, 0], [s, c, 0], [0, 0, 1.0]]).T + 1e-3 * rng.standard_normal((20000, 3))
    outs = []
    for search, refine in (("exact", "1"), ("grid", "1"), ("grid", "0")):
        monkeypatch.setenv("ICP_AMD_GRID_REFINE", refine)
        e = icp_amd.Engine(search=search)
        e.set_target(Q)
        e.set_source(P)
        e.iterate(4)
        outs.append((e.get_last()["idx"], e.get_source()))
        e.close()
    for o in outs[1:]:
        assert np.array_equal(outs[0][0], o[0]) and np.array_equal(outs[0][1], o[1])


def test_motion_certificate_changes_nothing(monkeypatch):
    """From the second grid iteration on a row keeps its match without visiting a cell when last iteration's search left a
    certificate (every other target at least D2 away) that is wider than the motion since (grid.hip).  It is only a
    shortcut: with it (default), without it (ICP_AMD_GRID_CERT=0) and through the brute-force search every iteration gives
    the same indices and, bit for bit, the same cloud -- on converging pairs (motion shrinks: almost every row skips), on a
    pair that starts far off (large motion: certificates fail), on tie lattices and duplicated targets (no certificate is
    ever issued) and across scales / offsets (the bound's rounding terms)."""
    rng = np.random.RandomState(5)
    cases = []
    P, Q = synth.make_pair(40000, 42000, seed=31)
    cases.append(("converging", P, Q, 14))
    P2, Q2 = synth.make_pair(20000, 20000, seed=32, angle=0.6, t0=(0.4, -0.3, 0.2))
    cases.append(("far", P2, Q2, 12))
    g = golden("lattice")
    cases.append(("lattice", g["P"], g["Q"], 8))
    Qd = np.concatenate([Q[:5000], Q[:5000][::-1], Q[5000:9000]])      # every one of the first 5000 targets twice
    cases.append(("duplicates", P[:6000], Qd, 8))
    for scale, shift in ((1e-4, 0.0), (1e5, 0.0), (1.0, 3e3), (1e-2, -40.0)):
        cases.append(("scaled %g %g" % (scale, shift), P[:15000] * scale + shift, Q[:16000] * scale + shift, 8))
    sheet = rng.uniform(-1, 1, (30000, 3)) * np.array([1.0, 1.0, 1e-3])
    cases.append(("sheet", sheet[:25000] @ synth.rotation(0.05).T + 0.01, sheet, 10))
    for name, Pc_, Qc_, iters in cases:
        runs = {}
        for label, search, cert in (("cert", "grid", "1"), ("nocert", "grid", "0"), ("brute", "f32", "1")):
            monkeypatch.setenv("ICP_AMD_GRID_CERT", cert)
            e = icp_amd.Engine(search=search)
            e.set_target(Qc_)
            e.set_source(Pc_)
            out = []
            for _ in range(iters):
                e.iterate(1)
                out.append(e.get_last()["idx"].copy())
            runs[label] = (out, e.get_source())
            e.close()
        monkeypatch.delenv("ICP_AMD_GRID_CERT")
        for other in ("nocert", "brute"):
            for it in range(iters):
                assert np.array_equal(runs["cert"][0][it], runs[other][0][it]), (name, other, it)
            assert np.array_equal(runs["cert"][1], runs[other][1]), (name, other)


def test_motion_certificate_is_void_after_two_updates_without_a_search():
    """The certificate describes ONE rigid motion; a caller that applies two updates between searches (step_update twice)
    must get a full search, not a shortcut priced for half the motion."""
    P, Q = synth.make_pair(30000, 30000, seed=33, angle=0.3)
    a = icp_amd.Engine(search="grid")
    b = icp_amd.Engine(search="f32")
    for e in (a, b):
        e.set_target(Q)
        e.set_source(P)
        e.step_begin()
        for _ in range(3):
            e.step_search()
            e.step_update()
        e.step_update()          # the same R, t applied once more, no search in between
        e.step_search()
        e.step_update()
        e.sync()
    assert np.array_equal(a.get_last()["idx"], b.get_last()["idx"])
    assert np.array_equal(a.get_source(), b.get_source())
    a.close()
    b.close()


def test_bound_from_the_visited_cells_with_targets_on_cell_edges(monkeypatch):
    """The bound on what a search has NOT visited counts the whole of the visited cells (grid.hip reach_lo / reach_hi): the
    coordinate next to the cells' outer edge is accepted only through cell_of().  Worst case for that: a target lattice whose
    points sit exactly ON the cell edges (17^3 integer points, 16^3 cells of edge exactly 1), queries on the edges, a hair
    off them (2^-20 and 2^-21 of a cell: the inset the code uses, and half of it), at cell centres and at random; the grid
    must return the oracle's index -- lowest index among exact ties -- stage by stage and through seeded engine iterations."""
    monkeypatch.setenv("ICP_AMD_GRID_PER_CELL", "1.2")
    monkeypatch.setenv("ICP_AMD_GRID_REFINE", "0")
    rng = np.random.RandomState(11)
    ax = np.arange(0, 17, dtype=np.float64)
    Q = np.stack(np.meshgrid(ax, ax, ax, indexing="ij"), -1).reshape(-1, 3)
    Q = Q[rng.permutation(len(Q))]
    Qz = Q - orc.mean0(Q)
    assert np.array_equal(orc.mean0(Q), np.full(3, 8.0))
    base = Qz[rng.permutation(len(Qz))[:1500]]
    parts = [base[:300]]
    for off in (0.5, 2.0 ** -20, -2.0 ** -20, 2.0 ** -21, -2.0 ** -21, 1.0 - 2.0 ** -20, 0.25, 1e-9):
        parts.append(base[:300] + off)
        parts.append(base[300:600] + np.array([off, 0.0, -off]))
        parts.append(base[600:900] + np.array([0.0, off, 0.5]))
    parts.append(rng.uniform(-9.5, 9.5, (3000, 3)))
    P = np.concatenate(parts)
    want = orc.correspondences(P, Qz)
    for search in ("grid", "exact"):
        assert np.array_equal(icp_amd.correspondences(P, np.zeros(3), Qz, search=search), want), search
    prev = np.roll(want, 3)          # a seed that is wrong for every row: the first box comes from the seed's distance
    assert np.array_equal(icp_amd.correspondences_seeded(P, np.zeros(3), Qz, prev, search="grid"), want)
    outs = []
    for search in ("grid", "exact"):
        e = icp_amd.Engine(search=search)
        e.set_target(Q)
        e.set_source(P[:6000] @ synth.rotation(0.01).T + 8.0 + np.array([0.02, -0.01, 0.015]))
        hist = []
        for _ in range(6):
            e.iterate(1)
            hist.append(e.get_last()["idx"].copy())
        outs.append((hist, e.get_source()))
        e.close()
    for it in range(6):
        assert np.array_equal(outs[0][0][it], outs[1][0][it]), it
    assert np.array_equal(outs[0][1], outs[1][1])


def test_processing_order_of_the_rows_changes_nothing(monkeypatch):
    """The grid search takes the rows in a spatial order while the cloud moves and in storage order once it has settled
    (ICP_AMD_GRID_ORDER_SWITCH, in cell half-edges of motion); rows that walk cells are re-packed inside their workgroup.
    Always spatial (0), storage order from the second search on (1e30), the default, no spatial order at all, and the
    centroid finished by its own launch instead of by the search must agree on every iteration's indices and on the final
    cloud, bit for bit."""
    P, Q = synth.make_pair(70000, 65000, seed=41, angle=0.2)
    runs = []
    for switch, order, fuse in (("0", "1", "1"), ("1e30", "1", "1"), (None, "1", "1"), (None, "0", "1"), (None, "1", "0")):
        if switch is None:
            monkeypatch.delenv("ICP_AMD_GRID_ORDER_SWITCH", raising=False)
        else:
            monkeypatch.setenv("ICP_AMD_GRID_ORDER_SWITCH", switch)
        monkeypatch.setenv("ICP_AMD_GRID_ORDER", order)
        monkeypatch.setenv("ICP_AMD_GRID_FUSE_CENTROID", fuse)   # 70 chunks: the search finishes the centroid itself, or not
        e = icp_amd.Engine(search="grid")
        e.set_target(Q)
        e.set_source(P)
        hist = []
        for _ in range(12):
            e.iterate(1)
            hist.append(e.get_last()["idx"].copy())
        runs.append((hist, e.get_source()))
        e.close()
    for r in runs[1:]:
        for it in range(12):
            assert np.array_equal(runs[0][0][it], r[0][it]), it
        assert np.array_equal(runs[0][1], r[1])


def test_settled_search_that_also_forms_the_covariance_changes_nothing(monkeypatch):
    """Once the previous search reports a settled registration the host launches search_grid_settled_kernel: one launch that
    searches (certificates first, the cells for the rows whose certificate fails) AND forms the chunk sums and indices
    cov_rec_kernel would.  Against the same engine with that launch switched off (ICP_AMD_GRID_FUSE_COV=0): indices, the
    covariance, R, t bit for bit in every iteration and the final cloud -- on a pair that converges (few rows walk), with the
    switch forced from the second seeded search on while the cloud still moves (most rows walk inside that kernel, and are read
    back by the lanes that own them), with the rejection hook on, on sizes that are no multiple of the chunk, on a small cloud
    whose search also finishes the centroid, and when the source cloud is downloaded / replaced in between (the SoA arrays are
    caught up from the records on demand)."""
    cases = [(synth.make_pair(70000, 65000, seed=43, angle=0.05), None, None),
             (synth.make_pair(70000, 65000, seed=44, angle=0.3), "1e30", None),      # "settled" from the start: rows walk in the fused launch
             (synth.make_pair(20011, 23000, seed=45, angle=0.2), "1e30", 0.05),      # + rejection of far pairs
             (synth.make_pair(3001, 2900, seed=46, angle=0.1), "1e30", None)]
    for (P, Q), switch, dmax in cases:
        runs = []
        for fuse in ("1", "0"):
            monkeypatch.setenv("ICP_AMD_GRID_FUSE_COV", fuse)
            if switch is None:
                monkeypatch.delenv("ICP_AMD_GRID_ORDER_SWITCH", raising=False)
            else:
                monkeypatch.setenv("ICP_AMD_GRID_ORDER_SWITCH", switch)
            e = icp_amd.Engine(search="grid")
            e.set_target(Q)
            e.set_source(P)
            if dmax:
                e.set_reject_distance(dmax)
            hist = []
            for it in range(14):
                e.iterate(1)
                last = e.get_last()
                hist.append((last["idx"].copy(), last["cov"].copy(), last["R"].copy(), last["t"].copy(), last["rms"]))
                if it == 6:
                    mid = e.get_source()          # catches the SoA arrays up; the records stay the running cloud
            runs.append((hist, mid, e.get_source(), e.last_motion()))
            e.close()
        a, b = runs
        for it in range(14):
            for x, y in zip(a[0][it], b[0][it]):
                assert np.array_equal(x, y), (len(P), switch, it)
        assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and a[3] == b[3]


def test_candidate_loads_in_flight_change_nothing(monkeypatch):
    """search_grid_kernel has three instantiations -- 4, 6 (default) or 8 candidate loads in flight per walking lane
    (ICP_AMD_DEBUG=grid_flight) -- that differ in how a lane's flattened candidate sequence is cut into batches (and in the masked
    slots at its end).  Same indices every iteration, same final cloud bit for bit, and the same as brute force: on a pair that
    starts far off (every row walks, boxes of many sizes), a thin sheet (long runs), duplicated targets and a tiny cloud."""
    rng = np.random.RandomState(9)
    cases = [synth.make_pair(30000, 31000, seed=71, angle=0.5, t0=(0.3, -0.2, 0.1))]
    sheet = rng.uniform(-1, 1, (20000, 3)) * np.array([1.0, 1.0, 1e-3])
    cases.append((sheet[:15000] @ synth.rotation(0.08).T + 0.02, sheet))
    P, Q = synth.make_pair(6000, 6000, seed=72, angle=0.2)
    cases.append((P, np.concatenate([Q, Q[:3000]])))
    cases.append(synth.make_pair(70, 1100, seed=73, angle=0.3))
    for P, Q in cases:
        runs = {}
        for label, search, debug in (("4", "grid", "grid_flight=4"), ("6", "grid", None), ("8", "grid", "grid_flight=8"), ("brute", "f32", None)):
            if debug:
                monkeypatch.setenv("ICP_AMD_DEBUG", debug)
            e = icp_amd.Engine(search=search)
            e.set_target(Q)
            e.set_source(P)
            out = []
            for _ in range(6):
                e.iterate(1)
                out.append(e.get_last()["idx"].copy())
            runs[label] = (out, e.get_source())
            e.close()
            if debug:
                monkeypatch.delenv("ICP_AMD_DEBUG")
        for other in ("4", "8", "brute"):
            for it in range(6):
                assert np.array_equal(runs["6"][0][it], runs[other][0][it]), (len(P), other, it)
            assert np.array_equal(runs["6"][1], runs[other][1]), (len(P), other)

