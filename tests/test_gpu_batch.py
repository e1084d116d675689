"""GPU: multi-registration mode (BASELINE config 5).  Every registration of a batch must equal the same pair run
alone through icp_amd_run, bit for bit, and match the oracle / golden vectors at the parity bar."""
import numpy as np
import pytest

import icp_amd
from conftest import golden, relerr
from icp_amd import synth
from oracle import pyoracle as orc

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("batch,n,m,iters", [(12, 1000, 1000, 6), (5, 700, 900, 4), (3, 1500, 3000, 3), (4, 64, 33, 3), (2, 2048, 4096, 2)])
def test_batch_equals_single_runs(batch, n, m, iters):
    Ps, Qs = [], []
    for b in range(batch):
        P, Q = synth.make_pair(n, max(n, m), seed=1000 + 7 * b, angle=0.05 + 0.01 * b)
        Ps.append(P)
        Qs.append(Q[:m] if m < len(Q) else Q)
    P, Q = np.stack(Ps), np.stack(Qs)
    out = icp_amd.icp_gpu_batch(P, Q, iters)
    for b in range(batch):
        P_one, tr = icp_amd.icp_gpu_optimized(P[b], Q[b], iters, trace=True)
        assert np.array_equal(out["idx"][b], tr["idx"][-1]), b
        assert np.array_equal(out["cov"][b], tr["cov"][-1]), b
        assert np.array_equal(out["R"][b], tr["R"][-1]) and np.array_equal(out["t"][b], tr["t"][-1]), b
        assert np.array_equal(out["P_out"][b], P_one), b
        assert out["rms"][b] == tr["rms"][-1]
    # and against the oracle for the first two registrations
    for b in range(min(batch, 2)):
        ref = orc.icp(P[b], Q[b], iters, trace=True, nthreads=4)
        assert np.array_equal(out["idx"][b], ref["idx"][-1])
        assert relerr(out["P_out"][b], ref["P_out"]) < 1e-9


def test_batch_of_golden_pairs():
    g1, g2 = golden("syn1k"), golden("syn256")
    for g in (g1, g2):
        iters = int(g["iters"])
        out = icp_amd.icp_gpu_batch(g["P"][None], g["Q"][None], iters)
        assert np.array_equal(out["idx"][0], g["idx"][-1])
        assert relerr(out["R"][0], g["R"][-1]) < 1e-9
        assert relerr(out["P_out"][0], g["P_out"]) < 1e-9


def test_batch_strict_order_carries_the_reference_bits(monkeypatch):
    """ICP_AMD_STRICT_ORDER=1 in the multi-registration kernel ("N <= 4096 and batch mode"): centroids and
    covariance summed row by row, one lane per output.  The first iteration's covariance is then BIT-identical to the
    reference's (src/cpu/icp.cpp:115-129 on centroids of src/libalg/mean.cpp:106-131) on every fixture the kernel can hold,
    every registration equals the same pair through a strict-order engine bit for bit at any iteration (both source size
    classes of the kernel, N != M), and without the switch the covariance is NOT the reference's everywhere."""
    for name in ("c1_30pts", "syn256", "syn1k"):
        g = golden(name)
        monkeypatch.setenv("ICP_AMD_STRICT_ORDER", "1")
        out = icp_amd.icp_gpu_batch(np.stack([g["P"], g["P"]]), np.stack([g["Q"], g["Q"]]), 1)
        for b in range(2):
            assert np.array_equal(out["cov"][b], g["cov"][0]), (name, b)
            assert np.array_equal(out["idx"][b], g["idx"][0]), (name, b)
        iters = int(g["iters"])
        out = icp_amd.icp_gpu_batch(g["P"][None], g["Q"][None], iters)
        monkeypatch.delenv("ICP_AMD_STRICT_ORDER")
        e = icp_amd.Engine(search="f32", strict_order=True)
        e.set_target(g["Q"])
        e.set_source(g["P"])
        e.iterate(iters)
        last = e.get_last()
        for k in ("cov", "R", "t", "idx"):
            assert np.array_equal(out[k][0], last[k]), (name, k)
        assert np.array_equal(out["P_out"][0], e.get_source()), name
        e.close()
        assert np.array_equal(out["idx"][0], g["idx"][-1]) and relerr(out["P_out"][0], g["P_out"]) < 1e-9
    differs = 0
    for n, m in ((1500, 3000), (2048, 4096), (700, 333)):
        P, Q = synth.make_pair(n, max(n, m), seed=61, angle=0.1)
        Q = Q[:m]
        monkeypatch.setenv("ICP_AMD_STRICT_ORDER", "1")
        out = icp_amd.icp_gpu_batch(np.stack([P, P[::-1]]), np.stack([Q, Q]), 3)
        monkeypatch.delenv("ICP_AMD_STRICT_ORDER")
        plain = icp_amd.icp_gpu_batch(P[None], Q[None], 3)
        differs += not np.array_equal(plain["cov"][0], out["cov"][0])
        assert relerr(plain["cov"][0], out["cov"][0]) < 1e-12 and np.array_equal(plain["idx"][0], out["idx"][0])
        for b, Pb in enumerate((P, P[::-1])):
            e = icp_amd.Engine(search="f32", strict_order=True)
            e.set_target(Q)
            e.set_source(np.ascontiguousarray(Pb))
            e.iterate(3)
            last = e.get_last()
            for k in ("cov", "R", "t", "idx"):
                assert np.array_equal(out[k][b], last[k]), (n, m, b, k)
            assert np.array_equal(out["P_out"][b], e.get_source())
            e.close()
    assert differs   # the default tree and the sequential sums do not round alike everywhere


def test_batch_variants_and_limits():
    P, Q = synth.make_batch(6, 500)
    b = icp_amd.Batch(6, 500, 500)
    b.upload(P, Q)
    b.run(5, search="exact")
    a = b.download()
    ms = b.run(5, search="fma", timed=True)
    c = b.download()
    b.run(5, search="auto")                 # the expanded form on {-2q, |q|^2} records (default)
    d = b.download()
    b.close()
    assert ms > 0
    for k in ("P_out", "idx", "R", "cov"):
        assert np.array_equal(a[k], c[k]), k
        assert np.array_equal(a[k], d[k]), k
    for shape in ((1, 2049, 100), (1, 100, 4097), (0, 10, 10), (1, 0, 10)):
        with pytest.raises(icp_amd.IcpAmdError):
            icp_amd.Batch(*shape)


def test_batch_expanded_form_is_only_a_filter():
    """Default batch search (expanded form): scales / offsets that make its error bound loose or unusable, exact ties
    (lattice, duplicated targets) and non-finite rows must give the EXACT variant's results bit for bit."""
    rng = np.random.RandomState(8)
    P, Q = synth.make_batch(4, 600)
    cases = [(P * s + o, Q * s + o) for s, o in ((1.0, 0.0), (1e-6, 0.0), (1e6, 0.0), (1.0, 1e4), (1e-3, 50.0), (1e20, 0.0),
                                                 (1e-20, 0.0), (1e150, 0.0), (1e-150, 0.0))]
    ax = np.arange(-4, 4, dtype=np.float64)
    L = np.stack(np.meshgrid(ax, ax, ax, indexing="ij"), -1).reshape(-1, 3)                     # 512 lattice points
    Lq = np.stack([L[rng.permutation(512)] for _ in range(4)])
    Lp = Lq[:, rng.permutation(512)] + np.array([0.5, 0.5, 0.0])                               # 4-way exact ties
    cases.append((Lp, Lq))
    Dq = Q.copy()
    Dq[:, 300:] = Dq[:, :300]                                                                   # duplicated targets
    cases.append((P, Dq))
    Pn = P.copy()
    Pn[0, 5, 0], Pn[1, 7, 1] = np.nan, np.inf
    cases.append((Pn, Q))
    for k, (Pc, Qc) in enumerate(cases):
        b = icp_amd.Batch(Pc.shape[0], Pc.shape[1], Qc.shape[1])
        b.upload(Pc, Qc)
        b.run(3, search="exact")
        want = b.download()
        b.run(3, search="auto")
        got = b.download()
        b.close()
        assert np.array_equal(want["idx"], got["idx"]), k
        assert np.array_equal(want["P_out"], got["P_out"], equal_nan=True), k


@pytest.mark.parametrize("seeded", ["1", "0"])
def test_batch_filters_on_ragged_target_counts(monkeypatch, seeded):
    """Target counts that are no multiple of the scan block (padding records sit inside the last block, and a block
    re-scan must not take them for points at the origin), fewer targets than one block, both source size classes, clouds
    around the origin: the default search (seeded from the second iteration on; ICP_AMD_BATCH_SEEDED=0 = the expanded form
    on every iteration) must give the EXACT variant's results bit for bit."""
    monkeypatch.setenv("ICP_AMD_BATCH_SEEDED", seeded)
    rng = np.random.RandomState(31)
    for n, m in ((1445, 95), (2048, 6), (121, 17), (1613, 3022), (1447, 1), (300, 7), (1024, 4095), (9, 2049)):
        for kind in range(3):
            if kind == 0:      # uniform around the origin
                Q, P = rng.uniform(-1, 1, (3, m, 3)), rng.uniform(-1, 1, (3, n, 3))
            elif kind == 1:    # small lattice with half-step offsets: exact ties, many duplicates
                Q = rng.randint(-5, 6, (3, m, 3)).astype(np.float64)
                P = rng.randint(-5, 6, (3, n, 3)).astype(np.float64) + rng.choice([0.0, 0.5], (3, n, 3))
            else:              # far from the origin at a small scale
                Q, P = rng.uniform(-1, 1, (3, m, 3)) * 1e-3 + 7.0, rng.uniform(-1, 1, (3, n, 3)) * 1e-3 + 7.0
            b = icp_amd.Batch(3, n, m)
            b.upload(P, Q)
            b.run(4, search="exact")
            want = b.download()
            b.run(4, search="auto")
            got = b.download()
            b.close()
            assert got["idx"].max() < m
            assert np.array_equal(want["idx"], got["idx"]), (n, m, kind)
            assert np.array_equal(want["P_out"], got["P_out"], equal_nan=True), (n, m, kind)


def test_batch_edge_shapes_and_iteration_counts():
    """One point, one target, the kernel's largest clouds, and 0 to 3 iterations (0: nothing runs; 1: only the first
    iteration's form; 2: the first seeded one): the default search against the EXACT variant, every output, bit for bit."""
    rng = np.random.RandomState(3)
    for n, m in ((1000, 1000), (1, 1), (2048, 4096), (5, 4096), (2048, 1)):
        P, Q = rng.uniform(-1, 1, (3, n, 3)), rng.uniform(-1, 1, (3, m, 3))
        b = icp_amd.Batch(3, n, m)
        b.upload(P, Q)
        for iters in (0, 1, 2, 3):
            b.run(iters, search="exact")
            want = b.download()
            b.run(iters, search="auto")
            got = b.download()
            for k in ("idx", "P_out", "R", "t", "cov", "rms"):
                assert np.array_equal(want[k], got[k], equal_nan=True), (n, m, iters, k)
        b.close()


def test_cli_batch_argument(tmp_path):
    import os
    import subprocess
    from conftest import ROOT
    P, Q = synth.make_batch(3, 200)
    for name, A in (("P.txt", P), ("Q.txt", Q)):
        with open(tmp_path / name, "w") as f:
            f.write("Points_0,Points_1,Points_2\n")   # without a header line the reference's reader drops the last row
            for r in A.reshape(-1, 3):
                f.write("%.17g,%.17g,%.17g\n" % tuple(r))
    exe = os.path.join(ROOT, "bin", "GPUICP")
    r = subprocess.run([exe, "P.txt", "Q.txt", "4", "-batch", "3"], cwd=tmp_path, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    lines = open(tmp_path / "point_cloud_result").read().split("\n")
    assert lines[0] == " dim0: 600 dim1: 3 array: "
    got = np.array([[float(v) for v in ln.split()] for ln in lines[1:601]])
    want = icp_amd.icp_gpu_batch(P, Q, 4)["P_out"].reshape(-1, 3)
    assert np.abs(got - want).max() < 1e-4
    r = subprocess.run([exe, "P.txt", "Q.txt", "4", "-batch", "7"], cwd=tmp_path, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "batch count must divide" in r.stderr


def test_run_batch_takes_clouds_of_any_size(tmp_path):
    """icp_amd_run_batch beyond the on-chip limits of the batch kernel (n <= 2048, m <= 4096): the registrations run through
    ordinary engines, several at a time on their own host threads -- each with exactly the results of icp_amd_run; and the
    executable's `-batch count` form accepts such files too."""
    import os
    import subprocess
    from conftest import ROOT
    rng = np.random.RandomState(3)
    b, n, m = 7, 3000, 5000
    Ps, Qs = [], []
    for k in range(b):
        P, Q = synth.make_pair(n, m, seed=300 + k)
        Ps.append(P); Qs.append(Q)
    P, Q = np.stack(Ps), np.stack(Qs)
    d = icp_amd.run_batch(P, Q, 4)
    for k in range(b):
        want, tr = icp_amd.icp_gpu_optimized(P[k], Q[k], 4, trace=True)
        assert np.array_equal(d["P_out"][k], want), k
        assert np.array_equal(d["idx"][k], tr["idx"][-1]) and np.array_equal(d["R"][k], tr["R"][-1]) and np.array_equal(d["t"][k], tr["t"][-1])
        assert d["rms"][k] == tr["rms"][-1]
    # small clouds still take the batch kernel
    Pb, Qb = synth.make_batch(5, 300)
    d2, d3 = icp_amd.run_batch(Pb, Qb, 3), icp_amd.icp_gpu_batch(Pb, Qb, 3)
    assert np.array_equal(d2["P_out"], d3["P_out"]) and np.array_equal(d2["idx"], d3["idx"])
    # the executable: two concatenated 2500-point clouds per file
    Pc, Qc = P[:2, :2500].reshape(-1, 3), Q[:2, :2500].reshape(-1, 3)
    for name, A in (("P.txt", Pc), ("Q.txt", Qc)):
        with open(tmp_path / name, "w") as f:
            f.write("Points_0,Points_1,Points_2\n")
            for r in A:
                f.write("%.17g,%.17g,%.17g\n" % tuple(r))
    r = subprocess.run([os.path.join(ROOT, "bin", "GPUICP"), "P.txt", "Q.txt", "3", "-batch", "2"], cwd=tmp_path, capture_output=True,
                       text=True, timeout=120, env=dict(os.environ, ICP_AMD_RESULT_FILE=str(tmp_path / "out.npy")))
    assert r.returncode == 0, r.stderr
    out, _ = icp_amd.read_cloud(tmp_path / "out.npy")
    for k in range(2):
        assert np.array_equal(out[k * 2500:(k + 1) * 2500], icp_amd.icp_gpu_optimized(Pc[k * 2500:(k + 1) * 2500], Qc[k * 2500:(k + 1) * 2500], 3)), k


def test_config5_full_size_4096_registrations_of_1k_points_20_iterations():
    """BASELINE configs[4] at its full size: 4096 registrations of 1000 x 1000 points, 20 iterations each, in ONE launch --
    the grid dimension the benchmark launches (bench.py --workload batch builds the same clouds: 64 distinct pairs, tiled).
    Every one of the 4096 results must equal the single-run result of its pair (icp_amd_run on that pair alone), bit for
    bit; two pairs are also held against the oracle.  Reference semantics for the record: the reference's own `-batch`
    (src/libgpuicp/batchcovs.cu:28-75) strip-mines ONE registration 16 source rows at a time -- it is not
    multi-registration, so there is no reference vector for this mode beyond the per-pair results."""
    batch, n, iters, distinct = 4096, 1000, 20, 64
    P64, Q64 = synth.make_batch(distinct, n, seed=synth.SEED_Q)
    reps = batch // distinct
    P, Q = np.tile(P64, (reps, 1, 1)), np.tile(Q64, (reps, 1, 1))
    out = icp_amd.icp_gpu_batch(P, Q, iters)
    for b in range(distinct):
        P_one, tr = icp_amd.icp_gpu_optimized(P64[b], Q64[b], iters, trace=True)
        for r in range(reps):                       # every copy of the pair, wherever it sits in the launch
            k = r * distinct + b
            assert np.array_equal(out["idx"][k], tr["idx"][-1]), k
            assert np.array_equal(out["P_out"][k], P_one), k
            assert np.array_equal(out["R"][k], tr["R"][-1]) and np.array_equal(out["t"][k], tr["t"][-1]), k
            assert np.array_equal(out["cov"][k], tr["cov"][-1]) and out["rms"][k] == tr["rms"][-1], k
    for b in (0, 37):
        ref = orc.icp(P64[b], Q64[b], iters, trace=True, nthreads=4)
        assert np.array_equal(out["idx"][b], ref["idx"][-1])
        assert relerr(out["R"][b], ref["R"][-1]) < 1e-9 and relerr(out["cov"][b], ref["cov"][-1]) < 1e-9
        assert relerr(out["P_out"][b], ref["P_out"]) < 1e-9
