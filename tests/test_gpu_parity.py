"""GPU parity tests (run on the MI355X box with -m gpu): the HIP path, called through the C ABI of
libicp_amd.so, against golden vectors from the unmodified reference and against the CPU oracle.
Bar: correspondence indices bit-exact; cov, R, t, P within 1e-9 relative (north_star), fp64."""
import os
import subprocess

import numpy as np
import pytest

import icp_amd
from conftest import ICP_CASES, ROOT, golden, relerr, relerr_t
from icp_amd import synth
from oracle import pyoracle as orc

pytestmark = pytest.mark.gpu
TOL = 1e-9


def test_device_fourth_root_is_correctly_rounded():
    """The comparison key sqrt(sqrt(d2)) must round exactly like the host's (IEEE correctly rounded sqrt)."""
    import ctypes as C
    rng = np.random.RandomState(1)
    x = np.concatenate([rng.uniform(0, 4, 400000), 10.0 ** rng.uniform(-300, 300, 400000),
                        np.array([0.0, 1.0, np.nextafter(1.0, 2.0), np.nextafter(1.0, 0.0), 5e-324, 1.7e308, np.inf])])
    # neighbourhoods of key-preimage edges: x = k^4 +- a few ulp
    k = rng.uniform(0.01, 3, 100000)
    e = (k * k) * (k * k)
    for s in range(-4, 5):
        x = np.concatenate([x, e * (1 + s * 2.0 ** -52)])
    out = np.empty_like(x)
    icp_amd.check(icp_amd.lib().icp_amd_test_fourth_root(x.ctypes.data_as(C.c_void_p), C.c_size_t(len(x)),
                                                            out.ctypes.data_as(C.c_void_p)))
    assert np.array_equal(out, np.sqrt(np.sqrt(x)))


@pytest.mark.parametrize("name", ICP_CASES)
def test_stagewise_against_reference(name):
    g = golden(name)
    Q, Qc = g["Q"], g["Qc"]
    assert relerr(icp_amd.centroid(Q), Qc) < 1e-13
    Qz = Q - Qc                                      # same IEEE subtraction as the reference's Q -= Q_center
    for slot, it in enumerate(g["P_in_iters"]):
        P_in, Pc = g["P_in"][slot], g["Pc"][it]
        assert relerr(icp_amd.centroid(P_in), Pc) < 1e-13
        for search, splits in (("exact", 1), ("fma", 1), ("f32", 1), ("exact", 3), ("fma", 5), ("f32", 2)):
            idx = icp_amd.correspondences(P_in, Pc, Qz, search=search, q_splits=splits)
            assert np.array_equal(idx, g["idx"][it]), (name, it, search, splits, int((idx != g["idx"][it]).sum()))
        cov = icp_amd.cross_cov(P_in, Pc, Qz, g["idx"][it])
        assert relerr(cov, g["cov"][it]) < 1e-12, (name, it)
        R, t, S = icp_amd.solve(g["cov"][it], Pc, Qc)
        assert relerr(R, g["R"][it]) < TOL and relerr(S, g["S"][it]) < TOL
        assert relerr_t(t, g["t"][it], Qc, Pc) < TOL
        P_next = icp_amd.apply(P_in, g["R"][it], g["t"][it])
        assert np.array_equal(P_next, orc.apply(P_in, g["R"][it], g["t"][it]))   # un-fused, same order: bit-exact


@pytest.mark.parametrize("name", ICP_CASES)
def test_whole_loop_against_reference(name):
    g = golden(name)
    iters = int(g["iters"])
    P_out, tr = icp_amd.icp_gpu_optimized(g["P"], g["Q"], iters, "-shared-loop", trace=True)
    mism = int((tr["idx"] != g["idx"]).sum())
    assert mism == 0, "%s: %d correspondence mismatches" % (name, mism)
    assert relerr(tr["Pc"], g["Pc"]) < TOL
    assert relerr(tr["cov"], g["cov"]) < TOL
    assert relerr(tr["R"], g["R"]) < TOL
    for it in range(iters):
        assert relerr_t(tr["t"][it], g["t"][it], g["Qc"], g["Pc"][it]) < TOL
    assert relerr(P_out, g["P_out"]) < TOL


def test_f32_prefilter_whole_loop_and_scale_robustness(monkeypatch):
    """The fp32 pre-filter is only a filter: whole-loop results equal the default path bit for bit, also for clouds
    whose scale / offset make the fp32 error bound loose (many slow-path visits) or unusable (falls back)."""
    monkeypatch.setenv("ICP_AMD_SEARCH", "f32")
    for name in ("cow_tr1", "syn3k_4k", "lattice", "c1_30pts"):
        g = golden(name)
        P_out, tr = icp_amd.icp_gpu_optimized(g["P"], g["Q"], int(g["iters"]), trace=True)
        assert np.array_equal(tr["idx"], g["idx"]), name
        assert relerr(P_out, g["P_out"]) < TOL
    monkeypatch.delenv("ICP_AMD_SEARCH")
    P, Q = synth.make_pair(5000, 6000, seed=77)
    for scale, shift in ((1.0, 0.0), (1e-6, 0.0), (1e6, 0.0), (1.0, 1e4), (1e-3, 50.0), (1e20, 0.0), (1e-20, 0.0)):
        Ps, Qs = P * scale + shift, Q * scale + shift
        a, ta = icp_amd.icp_gpu_optimized(Ps, Qs, 2, trace=True)
        monkeypatch.setenv("ICP_AMD_SEARCH", "f32")
        b, tb = icp_amd.icp_gpu_optimized(Ps, Qs, 2, trace=True)
        monkeypatch.delenv("ICP_AMD_SEARCH")
        assert np.array_equal(ta["idx"], tb["idx"]), (scale, shift)
        assert np.array_equal(a, b), (scale, shift)


def test_fp16_matrix_filter_stays_inside_its_error_bound():
    """The default F32 pre-filter computes sigma^2 (|q|^2 - 2 p.q) on v_mfma_f32_32x32x16_f16 from operands split into two
    fp16 pieces; the search is exact only if the value is within E' of the true one (search_mfma.hip).  A test kernel
    compares every filter value with the exact one (targets in runs of 32 identical points, so the instruction's result
    layout does not matter) and returns the worst |error| / E': at most 1 by the derivation -- in fact a small fraction of
    it -- over scales, offsets from the centre, coordinates near zero (lo pieces in fp16's subnormal range) and far sources."""
    import ctypes as C
    rng = np.random.RandomState(12)
    worst_seen = 0.0
    for case in range(14):
        scale = 10.0 ** rng.uniform(-12, 12) if case % 2 else 1.0
        T, n = 96, 700
        q = rng.uniform(-1, 1, (T, 3))
        p = rng.uniform(-1, 1, (n, 3))
        if case % 3 == 0:
            q[:, 2] = 0.0; p[::3, 1] = 0.0                      # exact zeros
        if case % 3 == 1:
            q[: T // 2] *= 10.0 ** rng.uniform(-9, -1, (T // 2, 1))   # coordinates tiny against qmax: lo pieces underflow
            p[: n // 2] *= 10.0 ** rng.uniform(-9, -1, (n // 2, 1))
        if case % 4 == 3:
            p[::5] *= rng.uniform(1, 100, (len(p[::5]), 1))      # sources well outside the target's extent (still in fp16 range)
        Qz = np.ascontiguousarray(np.repeat(q * scale, 32, axis=0))
        P = np.ascontiguousarray(p * scale)
        w = C.c_double()
        icp_amd.check(icp_amd.lib().icp_amd_test_mfma16_bound(P.ctypes.data_as(C.c_void_p), C.c_size_t(len(P)),
                                                              Qz.ctypes.data_as(C.c_void_p), C.c_size_t(len(Qz)), C.byref(w)))
        assert 0.0 <= w.value <= 1.0, (case, scale, w.value)
        worst_seen = max(worst_seen, w.value)
    assert worst_seen > 0.0          # the probe did compare something
    assert worst_seen < 0.5, worst_seen


@pytest.mark.parametrize("pipe,groups", [("0", "4"), ("16", "2"), ("16", "4")])
def test_matrix_pipe_prefilter_is_only_a_filter(monkeypatch, pipe, groups):
    """Both implementations of the F32 pre-filter: ICP_AMD_MFMA=16 (the default: v_mfma_f32_32x32x16_f16 on operands in two
    fp16 pieces, with 2 or 4 source groups per wave) and 0 (packed fp32 VALU).  Same indices as the oracle on the golden
    cases (whole loop, seeded iterations included), on the adversarial ties (stage level, several splits) and across
    scales / offsets, where its looser error bound only means more slow-path visits."""
    monkeypatch.setenv("ICP_AMD_SEARCH", "f32")
    monkeypatch.setenv("ICP_AMD_MFMA", pipe)
    monkeypatch.setenv("ICP_AMD_MFMA_G", groups)
    for name in ("cow_tr1", "syn3k_4k", "lattice", "c1_30pts"):
        g = golden(name)
        P_out, tr = icp_amd.icp_gpu_optimized(g["P"], g["Q"], int(g["iters"]), trace=True)
        assert np.array_equal(tr["idx"], g["idx"]), name
        assert relerr(P_out, g["P_out"]) < TOL
    g = golden("ties_corr")
    for splits in (1, 2, 3):
        idx = icp_amd.correspondences(g["Pz"], np.zeros(3), g["Qz"], search="f32", q_splits=splits)
        assert np.array_equal(idx, g["idx"]), splits
    P, Q = synth.make_pair(5000, 6000, seed=78)
    for scale, shift in ((1.0, 0.0), (1e-6, 0.0), (1e6, 0.0), (1.0, 1e4), (1e-3, 50.0), (1e20, 0.0)):
        Ps, Qs = P * scale + shift, Q * scale + shift
        b, tb = icp_amd.icp_gpu_optimized(Ps, Qs, 2, trace=True)
        monkeypatch.setenv("ICP_AMD_SEARCH", "exact")
        a, ta = icp_amd.icp_gpu_optimized(Ps, Qs, 2, trace=True)
        monkeypatch.setenv("ICP_AMD_SEARCH", "f32")
        assert np.array_equal(ta["idx"], tb["idx"]), (scale, shift)
        assert np.array_equal(a, b), (scale, shift)
    # sources far outside the range of the fp32 bound (the lane then takes the exact path on every block)
    Pf = P.copy()
    Pf[::7] *= 1e18
    b, tb = icp_amd.icp_gpu_optimized(Pf, Q, 1, trace=True)
    monkeypatch.setenv("ICP_AMD_SEARCH", "exact")
    a, ta = icp_amd.icp_gpu_optimized(Pf, Q, 1, trace=True)
    assert np.array_equal(ta["idx"], tb["idx"])


@pytest.mark.parametrize("method", ["-loop", "-shared", "-shared-loop", "-batch"])
def test_all_method_flags_give_the_same_result(method):
    g = golden("syn256")
    P_out = icp_amd.icp_gpu_optimized(g["P"], g["Q"], 3, method)
    ref = icp_amd.icp_gpu_optimized(g["P"], g["Q"], 3, "-shared-loop")
    assert np.array_equal(P_out, ref)
    with pytest.raises(icp_amd.IcpAmdError):
        icp_amd.icp_gpu_optimized(g["P"], g["Q"], 3, "-fast")


def test_adversarial_ties():
    g = golden("ties_corr")
    for search in ("exact", "fma", "f32"):
        for splits in (1, 2):
            idx = icp_amd.correspondences(g["Pz"], np.zeros(3), g["Qz"], search=search, q_splits=splits)
            assert np.array_equal(idx, g["idx"]), (search, splits)


def test_edge_shapes():
    rng = np.random.RandomState(3)
    for n, m in ((1, 1), (1, 5), (7, 1), (63, 65), (1025, 1023), (300, 2049), (2050, 257)):
        P, Q = rng.randn(n, 3), rng.randn(m, 3)
        Pc = orc.mean0(P)
        Qz = Q - orc.mean0(Q)
        want = orc.correspondences(P - Pc, Qz)
        for search in ("exact", "fma", "f32"):
            assert np.array_equal(icp_amd.correspondences(P, Pc, Qz, search=search, q_splits=2), want), (n, m, search)
        out = orc.icp(P, Q, 2, trace=True)
        P_out, tr = icp_amd.icp_gpu_optimized(P, Q, 2, trace=True)
        assert np.array_equal(tr["idx"], out["idx"])
        if n >= 7 and m >= 5:   # tiny clouds give singular covariances: only the cloud itself is compared there
            assert relerr(tr["R"], out["R"]) < TOL
        assert relerr(P_out, out["P_out"]) < 1e-8
    for P, Q in ((np.zeros((0, 3)), np.zeros((4, 3))), (np.zeros((4, 3)), np.zeros((0, 3)))):
        with pytest.raises(icp_amd.IcpAmdError):
            icp_amd.icp_gpu_optimized(P, Q, 1)
    with pytest.raises(icp_amd.IcpAmdError):
        icp_amd.icp_gpu_optimized(np.zeros((4, 2)), np.zeros((4, 3)), 1)


def test_duplicate_and_identical_clouds():
    """P == a permutation of Q: every point finds its own copy (index = inverse permutation), R = I, t = 0."""
    rng = np.random.RandomState(5)
    Q = rng.uniform(-1, 1, (5000, 3))
    perm = rng.permutation(5000)
    P = Q[perm]
    P_out, tr = icp_amd.icp_gpu_optimized(P, Q, 2, trace=True)
    assert np.array_equal(tr["idx"][0], perm)
    assert relerr(tr["R"][0], np.eye(3)) < 1e-12
    assert np.abs(P_out - P).max() < 1e-12


def test_full_size_properties_c3():
    """BASELINE config 3 size (100k x 100k): sampled rows against the oracle, determinism, and independence
    of the launch geometry (target splits, points per lane, kernel variant)."""
    P, Q = synth.make_pair(100000, 100000)
    results = []
    for kw in (dict(search="exact"), dict(search="fma"), dict(search="f32"), dict(search="exact", q_splits=7),
               dict(search="f32", q_splits=3), dict(search="exact")):
        e = icp_amd.Engine(**kw)
        e.set_target(Q)
        e.set_source(P)
        e.iterate(2)
        last = e.get_last()
        results.append((last, e.get_source()))
        Qc = e.target_centroid()
        e.close()
    base, P_base = results[0]
    for last, P_out in results[1:]:
        assert np.array_equal(last["idx"], base["idx"])
        assert np.array_equal(last["cov"], base["cov"])      # fixed-order reductions: bit-identical
        assert np.array_equal(P_out, P_base)
    # sampled oracle check of the first iteration
    e = icp_amd.Engine()
    e.set_target(Q)
    e.set_source(P)
    e.iterate(1)
    last = e.get_last()
    e.close()
    rows = np.random.RandomState(9).choice(100000, 1500, replace=False)
    Pc = orc.mean0(P)
    assert relerr(last["Pc"], Pc) < 1e-13
    want = orc.correspondences((P - last["Pc"])[rows], Q - Qc, nthreads=8)
    assert np.array_equal(last["idx"][rows], want)
    cov = orc.cross_cov(P - last["Pc"], Q - Qc, last["idx"])
    assert relerr(last["cov"], cov) < 1e-11
    # the registration makes progress on this pair: the matched-pair rms of iteration 2 is below iteration 1's
    assert base["rms"] < last["rms"]


def test_full_size_properties_c4_1m():
    """BASELINE config 4 size on one GPU (1M x 1M): sampled rows against the oracle, and all three search variants agree."""
    P, Q = synth.make_pair(1000000, 1000000)
    outs = {}
    for search in ("f32", "fma"):
        e = icp_amd.Engine(search=search)
        e.set_target(Q)
        e.set_source(P)
        e.iterate(2)
        outs[search] = (e.get_last(), e.get_source(), e.target_centroid())
        e.close()
    a, b = outs["f32"], outs["fma"]
    assert np.array_equal(a[0]["idx"], b[0]["idx"]) and np.array_equal(a[0]["cov"], b[0]["cov"]) and np.array_equal(a[1], b[1])
    e = icp_amd.Engine()
    e.set_target(Q)
    e.set_source(P)
    e.iterate(1)
    last = e.get_last()
    e.close()
    rows = np.random.RandomState(3).choice(1000000, 400, replace=False)
    want = orc.correspondences((P - last["Pc"])[rows], Q - a[2], nthreads=16)
    assert np.array_equal(last["idx"][rows], want)
    assert relerr(last["Pc"], orc.mean0(P)) < 1e-13
    assert relerr(last["cov"], orc.cross_cov(P - last["Pc"], Q - a[2], last["idx"])) < 1e-11


def test_target_can_be_replaced_under_an_existing_source():
    """set_target after set_source re-plans the search (variant, splits) for the new target."""
    P, Q1 = synth.make_pair(3000, 5000, seed=31)
    _, Q2 = synth.make_pair(3000, 2500, seed=32)
    e = icp_amd.Engine()
    e.set_target(Q1)
    e.set_source(P)
    e.iterate(1)
    e.set_source(P * 1e25)
    e.set_target(Q2 * 1e25)          # outside the fp32 pre-filter's range: must fall back, not mis-filter
    e.iterate(2)
    got = e.get_last()["idx"]
    e.close()
    ref = orc.icp(P * 1e25, Q2 * 1e25, 2, trace=True, nthreads=4)
    assert np.array_equal(got, ref["idx"][-1])


def test_init_is_optional_and_repeatable():
    """icp_amd_init (context + code objects up front, for callers that overlap it with their I/O): fine before, between
    and after other calls, from another thread too; a device that does not exist is an argument error."""
    import threading
    icp_amd.init()
    t = threading.Thread(target=icp_amd.init, args=(0,))
    t.start(); t.join()
    g = golden("syn256")
    a = icp_amd.icp_gpu_optimized(g["P"], g["Q"], 2)
    icp_amd.init(0)
    assert np.array_equal(icp_amd.icp_gpu_optimized(g["P"], g["Q"], 2), a)
    with pytest.raises(icp_amd.IcpAmdError):
        icp_amd.init(icp_amd.device_count())


def test_released_device_memory_is_reused_and_can_be_trimmed():
    """devmem.cpp: blocks released by a finished registration are kept and handed to the next one (dirty, so this is also a
    test that nothing relies on fresh zeroed memory); icp_amd_cache_trim gives them back.  Results do not change."""
    g = golden("syn3k_4k")
    icp_amd.cache_trim()
    a, ta = icp_amd.icp_gpu_optimized(g["P"], g["Q"], 3, trace=True)
    held = icp_amd.cache_trim()
    assert held > 3 * 8 * (len(g["P"]) + len(g["Q"]))          # at least the two clouds' worth of blocks was being kept
    assert icp_amd.cache_trim() == 0
    for _ in range(3):                                           # second and third run on recycled blocks
        b, tb = icp_amd.icp_gpu_optimized(g["P"], g["Q"], 3, trace=True)
        assert np.array_equal(a, b) and np.array_equal(ta["idx"], tb["idx"]) and np.array_equal(ta["cov"], tb["cov"])
    # a different size in between, then the first size again
    h = golden("cow_tr1")
    icp_amd.icp_gpu_optimized(h["P"], h["Q"], 2)
    b = icp_amd.icp_gpu_optimized(g["P"], g["Q"], 3)
    assert np.array_equal(a, b)
    assert icp_amd.cache_trim() > 0


def test_nothing_relies_on_fresh_or_stale_device_memory():
    """ICP_AMD_CACHE_POISON=1 hands every device block out filled with the 32-bit pattern 3 (a small valid-looking index, a
    denormal as a double).  A few whole-path cases in a child process: all variants, a replaced target, batch mode."""
    import sys
    code = (
        "import numpy as np, icp_amd, sys; sys.path.insert(0, %r); from conftest import golden; from icp_amd import synth\n"
        "for name in ('cow_tr1', 'syn3k_4k', 'lattice'):\n"
        "    g = golden(name)\n"
        "    for s in ('auto', 'f32', 'fma', 'exact', 'grid'):\n"
        "        e = icp_amd.Engine(search=s); e.set_target(g['Q']); e.set_source(g['P']); e.iterate(int(g['iters']))\n"
        "        assert np.array_equal(e.get_last()['idx'], g['idx'][-1]), (name, s)\n"
        "        e.set_target(g['Q'][: len(g['Q']) // 2]); e.iterate(1); e.close()\n"
        "P, Q = synth.make_batch(4, 500)\n"
        "a = icp_amd.icp_gpu_batch(P, Q, 3)['P_out']\n"
        "for k in range(4): assert np.array_equal(a[k], icp_amd.icp_gpu_optimized(P[k], Q[k], 3)), k\n"
        "print('poison ok')\n" % os.path.join(ROOT, "tests"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=200,
                       env=dict(os.environ, ICP_AMD_CACHE_POISON="1", PYTHONPATH=ROOT))
    assert r.returncode == 0 and "poison ok" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def test_concurrent_registrations_from_several_threads():
    """The library keeps no hidden global state between engines: registrations running at the same time on four host
    threads (ctypes releases the GIL; each call has its own engine and stream, they share the device-memory cache and the
    thread-local error text) return what they return one after the other."""
    import threading
    cases = [golden(n) for n in ("cow_tr1", "syn3k_4k", "syn1k", "lattice")]
    want = [icp_amd.icp_gpu_optimized(g["P"], g["Q"], 4) for g in cases]
    got, errs = {}, []
    def work(k, rounds):
        try:
            for r in range(rounds):
                g = cases[(k + r) % len(cases)]
                got[(k, r)] = ((k + r) % len(cases), icp_amd.icp_gpu_optimized(g["P"], g["Q"], 4))
            with pytest.raises(icp_amd.IcpAmdError) as ei:      # an error in one thread does not leak into another's text
                icp_amd.icp_gpu_optimized(np.zeros((0, 3)), cases[0]["Q"], 1)
            assert "empty" in str(ei.value) or "NULL" in str(ei.value) or str(ei.value)
        except Exception as exc:   # noqa: BLE001 -- reported below, in the main thread
            errs.append(repr(exc))
    threads = [threading.Thread(target=work, args=(k, 6)) for k in range(4)]
    for t in threads: t.start()
    for t in threads: t.join()
    assert not errs, errs
    assert len(got) == 24
    for (k, r), (c, out) in got.items():
        assert np.array_equal(out, want[c]), (k, r, c)


def test_cli_matches_reference_output(tmp_path):
    """GPUICP on the 30-point pair: same stdout line shape, same output file format as the reference
    (reference src/gpu/main.cu:79-80); usage and method errors abort like runtime_assert."""
    exe = os.path.join(ROOT, "bin", "GPUICP")
    g = golden("c1_30pts")
    pf, qf = tmp_path / "P.txt", tmp_path / "Q.txt"
    for path, A in ((pf, g["P"]), (qf, g["Q"])):
        with open(path, "w") as f:
            f.write("Points_0,Points_1,Points_2\n")
            for r in A:
                f.write("%.17g,%.17g,%.17g\n" % tuple(r))
    r = subprocess.run([exe, str(pf), str(qf), "5", "-shared-loop"], cwd=tmp_path, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert r.stdout.startswith("Squared actual mean diff: ")
    assert float(r.stdout.split(":")[1]) < 1e-12
    lines = open(tmp_path / "point_cloud_result").read().split("\n")
    assert lines[0] == " dim0: 30 dim1: 3 array: "
    got = np.array([[float(v) for v in ln.split()] for ln in lines[1:31]])
    assert np.abs(got - g["P_out"]).max() < 1e-4      # the file carries 6 significant digits
    r = subprocess.run([exe, str(pf), str(qf), "5"], cwd=tmp_path, capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "Usage: ./GPUICP file1 file2 nbiters method" in r.stderr
    r = subprocess.run([exe, str(pf), str(qf), "5", "-fast"], cwd=tmp_path, capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "Invalid method, must be: -shared|-loop|-shared-loop|-batch" in r.stderr


def test_cli_five_column_cloud_aborts_like_the_reference(tmp_path):
    """data/icp/cow_tr2.txt has FIVE columns (Points_0..2, Points_Magnitude, Point ID).  The reference reads it and aborts
    with an uncaught runtime_error -- `element_wise_reduce requires arrays to be broadcastable` (src/libalg/alg.cpp:21;
    observed on oracle/_ref/CPUICP cow_tr2 cow_ref 5: that text on stderr, SIGABRT) -- and GPUICP does the same: same text,
    abort status, no result file, nothing on stdout."""
    exe = os.path.join(ROOT, "bin", "GPUICP")
    g = golden("cow_tr1")
    P, Q = g["P"], g["Q"]
    with open(tmp_path / "cow_tr2.txt", "w") as f:
        f.write("Points_0,Points_1,Points_2,Points_Magnitude,Point ID\n")
        for k, r in enumerate(P):
            f.write("%.6g,%.6g,%.6g,%.6g,%d\n" % (r[0], r[1], r[2], float(np.sqrt((r * r).sum())), k))
    with open(tmp_path / "cow_ref.txt", "w") as f:
        f.write("Points_0,Points_1,Points_2\n")
        for r in Q:
            f.write("%.17g,%.17g,%.17g\n" % tuple(r))
    A, hdr = icp_amd.read_csv(tmp_path / "cow_tr2.txt")
    assert A.shape == (len(P), 5) and hdr.startswith("Points_0,Points_1,Points_2,Points_Magnitude,Point ID")
    for method in ("-shared-loop", "-batch"):
        r = subprocess.run([exe, "cow_tr2.txt", "cow_ref.txt", "5", method], cwd=tmp_path, capture_output=True, text=True, timeout=120)
        assert r.returncode in (-6, 134), (r.returncode, r.stderr)
        assert "element_wise_reduce requires arrays to be broadcastable" in r.stderr and r.stdout == ""
        assert not os.path.exists(tmp_path / "point_cloud_result")
    # the other way round (5-column target) ends the same way
    r = subprocess.run([exe, "cow_ref.txt", "cow_tr2.txt", "5", "-loop"], cwd=tmp_path, capture_output=True, text=True, timeout=120)
    assert r.returncode in (-6, 134) and "element_wise_reduce requires arrays to be broadcastable" in r.stderr


def test_device_svd_against_numpy_on_random_and_degenerate_covariances():
    """icp_amd_solve (the one-sided Jacobi of kernels.hip) against numpy.linalg.svd -- LAPACK gesdd, an algorithm that shares
    nothing with the device's or the oracle's Jacobi -- on covariances no fixture holds: random (scales 1e-286 ... 1e286), ill-conditioned,
    negative-determinant (R = U V^T is then a reflection: the reference has no det fix, src/cpu/icp.cpp:218-221), rank 2 and
    rank 1.  S within 1e-9 of S0; R = U V^T within 1e-9 wherever it is unique (S2 > 1e-6 S0); for rank-2 input R is unique up
    to the sign of u3 v3^T; R is orthogonal in every case and t = Qc - Pc R^T."""
    rng = np.random.RandomState(77)
    cases = []
    for k in range(220):
        U, _ = np.linalg.qr(rng.standard_normal((3, 3)))
        V, _ = np.linalg.qr(rng.standard_normal((3, 3)))
        kind = k % 5
        if kind == 0:   # any scale: the device scales the matrix by a power of two first, so no square under- or overflows
            A = rng.standard_normal((3, 3)) * 10.0 ** rng.uniform(-6, 6) * (10.0 ** rng.choice([-280, -150, 0, 0, 150, 280]) if k % 10 == 0 else 1.0)
        elif kind == 1:   # ill-conditioned but nonsingular
            A = U @ np.diag([1.0, 10.0 ** rng.uniform(-5, 0), 10.0 ** rng.uniform(-5.5, -5)]) @ V.T * 10.0 ** rng.uniform(-3, 3)
        elif kind == 2:   # negative determinant
            A = rng.standard_normal((3, 3))
            if np.linalg.det(A) > 0:
                A[0] = -A[0]
            A *= 10.0 ** rng.uniform(-2, 4)
        elif kind == 3:   # rank 2 (a planar cloud in general position)
            A = U @ np.diag([rng.uniform(1, 5), rng.uniform(0.1, 1), 0.0]) @ V.T
        else:             # rank 1
            A = np.outer(U[:, 0], V[:, 0]) * rng.uniform(0.5, 50)
        cases.append((kind, A))
    worst = 0.0
    for kind, A in cases:
        Pc, Qc = rng.uniform(-3, 3, 3), rng.uniform(-3, 3, 3)
        R, t, S = icp_amd.solve(A, Pc, Qc)
        Un, Sn, Vtn = np.linalg.svd(A)
        assert np.abs(S - Sn).max() <= 1e-9 * Sn[0], (kind, S, Sn)
        assert np.abs(R.T @ R - np.eye(3)).max() < 1e-12, kind                     # orthogonal, reflection or not
        assert np.abs(t - (Qc - R @ Pc)).max() < 1e-12 * max(1.0, np.abs(Pc).max(), np.abs(Qc).max())
        Rn = Un @ Vtn
        if Sn[2] > 1e-6 * Sn[0]:
            err = np.abs(R - Rn).max()
            assert err < TOL, (kind, err, Sn)
            assert np.sign(np.linalg.det(R)) == np.sign(np.linalg.det(A / np.abs(A).max()))   # no reflection fix (det of the normalised matrix: 1e-146^3 underflows)
            worst = max(worst, err)
        elif Sn[1] > 1e-6 * Sn[0]:
            # rank 2: u3, v3 are determined up to one sign each, so U V^T is one of two matrices
            flip = Rn - 2.0 * np.outer(Un[:, 2], Vtn[2])
            assert min(np.abs(R - Rn).max(), np.abs(R - flip).max()) < 1e-8, (kind, Sn)
    assert worst < TOL


@pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "CPUICP")), reason="needs the built oracle/_ref/CPUICP")
def test_cli_against_the_reference_executable_on_random_pairs(tmp_path):
    """The two executables side by side on inputs no fixture holds: oracle/_ref/CPUICP (the reference's src/cpu/main.cpp,
    compiled unmodified; CPU) and bin/GPUICP (this library; GPU) read the same CSV files.  The result files agree -- same
    first line, every value within the 6 digits the format carries -- and so does the printed report.  One pair is
    header-less: both programs lose its last row (csv.cpp:147-148)."""
    exe, ref = os.path.join(ROOT, "bin", "GPUICP"), os.path.join(ROOT, "oracle", "_ref", "CPUICP")
    rng = np.random.RandomState(21)
    def parse(path):
        lines = open(path).read().split("\n")
        return lines[0], np.array([[float(v) for v in ln.split()] for ln in lines[1:] if ln.strip()])
    for case in range(10):
        n = int(rng.randint(40, 1500))
        scale = 10.0 ** rng.uniform(-3, 3)
        Q = rng.uniform(-1, 1, (n, 3)) * scale
        a = rng.uniform(-0.4, 0.4)
        R = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])
        P = Q[rng.permutation(n)] @ R.T + rng.uniform(-0.1, 0.1, 3) * scale + rng.standard_normal((n, 3)) * scale * 1e-3
        header = case != 3
        for name, A in (("P.txt", P), ("Q.txt", Q)):
            with open(tmp_path / name, "w") as f:
                if header:
                    f.write("Points_0,Points_1,Points_2\n")
                for r in A:
                    f.write("%.17g,%.17g,%.17g\n" % tuple(r))
        iters = int(rng.randint(1, 7))
        outs = []
        for cmd, sub in (([ref, "P.txt", "Q.txt", str(iters)], "r"), ([exe, "P.txt", "Q.txt", str(iters), "-shared-loop"], "g")):
            d = tmp_path / ("%s%d" % (sub, case))
            d.mkdir()
            for name in ("P.txt", "Q.txt"):
                os.link(tmp_path / name, d / name)
            r = subprocess.run(cmd, cwd=d, capture_output=True, text=True, timeout=200)
            assert r.returncode == 0, (case, cmd[0], r.stderr[-2000:])
            outs.append((r.stdout.strip(), parse(d / "point_cloud_result")))
        (line_r, (h_r, A_r)), (line_g, (h_g, A_g)) = outs
        assert h_r == h_g == " dim0: %d dim1: 3 array: " % (n if header else n - 1), case
        assert A_r.shape == A_g.shape
        assert np.abs(A_r - A_g).max() <= 2e-5 * max(np.abs(A_r).max(), 1e-300), case
        d_r, d_g = float(line_r.split(":")[1]), float(line_g.split(":")[1])
        assert line_r.startswith("Squared actual mean diff: ") and line_g.startswith("Squared actual mean diff: ")
        assert abs(d_r - d_g) <= 2e-5 * max(abs(d_r), 1e-12 * scale), (case, line_r, line_g)


def test_stage_level_f32_fallback_uses_the_requested_splits():
    """Regression (found by tests/fuzz/fuzz_variants.py): a target whose range defeats the fp32 bound makes the stage-level
    F32 search fall back to FMA; it must run with the same number of target splits the merge then reads."""
    rng = np.random.RandomState(4)
    for n, m, scale in ((2799, 142, 1e22), (1768, 796, 1e-25), (128, 3196, 1e17)):
        P, Q = rng.uniform(-1, 1, (n, 3)) * scale, rng.uniform(-1, 1, (m, 3)) * scale
        Pc, Qz = orc.mean0(P), Q - orc.mean0(Q)
        want = orc.correspondences(P - Pc, Qz, nthreads=4)
        for splits in (1, 3, 6, 8):
            assert np.array_equal(icp_amd.correspondences(P, Pc, Qz, search="f32", q_splits=splits), want), (n, m, scale, splits)


def test_differential_fuzz_of_all_variants_short():
    """A few seconds of tests/fuzz/fuzz_variants.py: lattices, duplicates, clusters, few-ulp near-ties, wild scales, flat clouds;
    every search variant, stage level and seeded engine iterations, against the oracle."""
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz", "fuzz_variants.py"), "8", "11"], capture_output=True, text=True,
                       timeout=200)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "mismatches 0" in r.stdout


def test_cli_on_the_cow_pair_20_iterations(tmp_path):
    """BASELINE config 2 through the executable: `GPUICP cow_tr1 cow_ref 20 -shared-loop` (inputs from the golden
    fixture, written as CSV with the reference's header) reproduces the reference CPUICP result file and its report."""
    exe = os.path.join(ROOT, "bin", "GPUICP")
    g = golden("cow_tr1")
    for name, A in (("P.txt", g["P"]), ("Q.txt", g["Q"])):
        with open(tmp_path / name, "w") as f:
            f.write("Points_0,Points_1,Points_2\n")
            for r in A:
                f.write("%.17g,%.17g,%.17g\n" % tuple(r))
    env = dict(os.environ, ICP_AMD_RESULT_FILE=str(tmp_path / "out.npy"))
    r = subprocess.run([exe, "P.txt", "Q.txt", str(int(g["iters"])), "-shared-loop"], cwd=tmp_path, capture_output=True, text=True,
                       timeout=120, env=env)
    assert r.returncode == 0, r.stderr
    want_report = orc.euclidian_distance(g["Q"], g["P_out"])
    assert abs(float(r.stdout.split(":")[1]) - want_report) <= 1e-6 * max(1.0, want_report)   # printed with 6 digits
    assert relerr(np.load(tmp_path / "out.npy"), g["P_out"]) < TOL
    lines = open(tmp_path / "point_cloud_result").read().split("\n")
    assert lines[0] == " dim0: %d dim1: 3 array: " % len(g["P"])
    # --bench (additive sixth argument): the same stdout line and result file, one JSON line on stderr
    import json
    first_out, first_file = r.stdout, open(tmp_path / "point_cloud_result").read()
    rb = subprocess.run([exe, "P.txt", "Q.txt", str(int(g["iters"])), "-shared-loop", "--bench"], cwd=tmp_path, capture_output=True,
                        text=True, timeout=120, env=env)
    assert rb.returncode == 0, rb.stderr
    assert rb.stdout == first_out and open(tmp_path / "point_cloud_result").read() == first_file
    d = json.loads([ln for ln in rb.stderr.splitlines() if ln.startswith("{")][-1])
    assert d["n"] == d["m"] == len(g["P"]) and d["iters"] == int(g["iters"]) and d["gpus"] == 1 and d["cpu_cores"] >= 1
    assert d["iters_per_sec"] > 0 and d["pairs_per_sec"] > 0 and d["search"]["variant"] == "grid" and d["search"]["launches"] == d["iters"]
    assert 0 < d["roofline"]["frac"] < 1 and d["roofline"]["bound"] == "hbm"


@pytest.mark.gpu
@pytest.mark.parametrize("matrix_pipe", ["16", "0"])
def test_filter_variants_nonfinite_and_outliers(matrix_pipe, monkeypatch):
    """Non-finite and far-out-of-range rows on either side: the filtered variants (fp16 matrix pipe, packed fp32) give
    the exact variant's correspondences, which are the oracle's.  A non-finite or huge target drives the filter's scale
    and bound to their limits (every block then takes the exact path); a non-finite source row has no correspondence."""
    monkeypatch.setenv("ICP_AMD_MFMA", matrix_pipe)
    P, Q = synth.make_pair(3000, 4100, seed=79)
    Pz, Qz = P - orc.mean0(P), Q - orc.mean0(Q)
    cases = {}
    Pn = Pz.copy()
    Pn[5, 0], Pn[17, 1], Pn[99, 2], Pn[640] = np.nan, np.inf, -np.inf, 1e200
    cases["source"] = (Pn, Qz)
    # source rows far outside the filters' range whose exact d^2 is still finite: the fp32 products p x (-2q) overflow
    # (|p| 2|q| > 3.4e38) although the row has a perfectly good nearest neighbour
    Pf = Pz.copy()
    Pf[3], Pf[70, 1], Pf[1234, 2], Pf[2999] = 1e30, -3e38, 7e60, (1e100, -1e100, 1e100)
    cases["source_far_finite"] = (Pf, Qz * 1e9)
    cases["source_far_finite_big_targets"] = (Pf, Qz * 1e14)
    for name, vals in (("target_inf", (np.inf, np.nan)), ("target_huge", (1e19, -3e25)), ("target_tiny", (1e-300, 5e-324))):
        Qn = Qz.copy()
        Qn[7, 1], Qn[4000, 0] = vals
        cases[name] = (Pz, Qn)
    Qs = Qz * 1e-200
    cases["all_tiny"] = (Pz * 1e-200, Qs)
    cases["all_huge"] = (Pz * 1e150, Qz * 1e150)
    for name, (a, b) in cases.items():
        want = icp_amd.correspondences(a, np.zeros(3), b, search="exact")
        if name == "source":
            assert want[5] == -1 and want[17] == -1 and want[99] == -1
        if np.isfinite(a).all() and np.isfinite(b).all():
            assert np.array_equal(want, orc.correspondences(a, b)), name
        for search in ("fma", "f32"):
            for splits in (1, 3):
                got = icp_amd.correspondences(a, np.zeros(3), b, search=search, q_splits=splits)
                assert np.array_equal(got, want), (name, search, splits)


def test_debug_switches_come_from_one_list(monkeypatch):
    """Every debugging / test switch is an entry of ICP_AMD_DEBUG="name=value,..."; the older spelling as a variable of its own
    (ICP_AMD_<NAME>) is still read, the list wins, and names that merely share a prefix do not match."""
    P, Q = synth.make_pair(5000, 6000, seed=12)
    def splits():
        e = icp_amd.Engine(search="f32")
        e.set_target(Q)
        e.set_source(P)
        geo = e.search_geometry()
        e.close()
        return geo["q_splits"]
    monkeypatch.delenv("ICP_AMD_DEBUG", raising=False)
    monkeypatch.delenv("ICP_AMD_QSPLITS", raising=False)
    default = splits()
    monkeypatch.setenv("ICP_AMD_DEBUG", "grid_cert=0, qsplits=5,mfma_g=2")
    assert splits() == 5
    monkeypatch.setenv("ICP_AMD_QSPLITS", "7")
    assert splits() == 5                                  # the list wins
    monkeypatch.setenv("ICP_AMD_DEBUG", "qsplitsx=3,xqsplits=4")
    assert splits() == 7                                  # near-misses in the list are not this switch; the old spelling is
    monkeypatch.delenv("ICP_AMD_QSPLITS")
    assert splits() == default


@pytest.mark.parametrize("name", ["c1_30pts", "syn256", "syn1k", "cow_tr1"])
def test_strict_order_reductions_are_bit_identical_to_the_reference(name, monkeypatch):
    """ICP_AMD_STRICT_ORDER=1 / Engine(strict_order=True) (clouds of up to 4096 rows): the centroid as the
    reference's sequential sum over the rows followed by one division (src/libalg/mean.cpp:106-131) and the cross-covariance as
    its row-by-row cov += q_a p_b (src/cpu/icp.cpp:115-129), one lane per output: Qc, Pc and cov are then BIT-identical to the
    reference's on the reference's own running clouds, where the default trees agree to ~1e-15.  Indices do not change."""
    g = golden(name)
    P, Q, Qc = g["P"], g["Q"], g["Qc"]
    monkeypatch.setenv("ICP_AMD_STRICT_ORDER", "1")
    assert np.array_equal(icp_amd.centroid(Q), Qc)
    Qz = Q - Qc
    for slot, it in enumerate(g["P_in_iters"]):
        P_in = g["P_in"][slot]
        assert np.array_equal(icp_amd.centroid(P_in), g["Pc"][it]), (name, it)
        assert np.array_equal(icp_amd.cross_cov(P_in, g["Pc"][it], Qz, g["idx"][it]), g["cov"][it]), (name, it)
    monkeypatch.delenv("ICP_AMD_STRICT_ORDER")
    # the engine: first iteration bit for bit (later ones start from a cloud that went through our SVD instead of LAPACK's)
    for search in ("auto", "f32"):
        e = icp_amd.Engine(search=search, strict_order=True)
        e.set_target(Q)
        e.set_source(P)
        assert np.array_equal(e.target_centroid(), Qc)
        e.iterate(1)
        last = e.get_last()
        assert np.array_equal(last["Pc"], g["Pc"][0]) and np.array_equal(last["cov"], g["cov"][0]), (name, search)
        assert np.array_equal(last["idx"], g["idx"][0])
        e.iterate(int(g["iters"]) - 1)
        last = e.get_last()
        assert np.array_equal(last["idx"], g["idx"][-1])
        assert relerr(last["R"], g["R"][-1]) < TOL and relerr(e.get_source(), g["P_out"]) < TOL
        e.close()
    # and the default order is NOT bit-identical everywhere (else this test would prove nothing) but agrees to 1e-13
    d = icp_amd.centroid(g["P_in"][0])
    assert relerr(d, g["Pc"][0]) < 1e-13
