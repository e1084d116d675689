"""CPU tests (no GPU): the C-ABI library loads and exports every declared symbol, the host-side pieces kept
from the reference (CSV loader, writer, final report) behave like the reference, and the compute entry
points fail loudly without a device (no CPU fallback)."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import icp_amd
from conftest import ROOT, golden
from oracle import pyoracle as orc

REF_DATA = "/root/reference/data"


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "icp_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b(icp_amd_[a-z_0-9]+)\s*\(", hdr))
    assert len(names) >= 30
    L = icp_amd.lib()
    for n in sorted(names):
        assert hasattr(L, n), "libicp_amd.so does not export %s" % n


def test_no_cpu_fallback():
    if icp_amd.device_count() > 0:
        pytest.skip("a GPU is present")
    P = np.zeros((4, 3))
    for fn in (lambda: icp_amd.icp_gpu_optimized(P, P, 1), lambda: icp_amd.centroid(P),
               lambda: icp_amd.correspondences(P, np.zeros(3), P), lambda: icp_amd.Engine(), lambda: icp_amd.init()):
        with pytest.raises(icp_amd.IcpAmdError) as ei:
            fn()
        assert ei.value.code == -2 and "no CPU fallback" in str(ei.value)


def test_sharding_arithmetic():
    for n, g in ((1000000, 8), (1000000, 1), (1048576, 8), (100000, 4), (1, 8), (1025, 2)):
        rpr, cpr = icp_amd.rows_per_rank(n, g), icp_amd.chunks_per_rank(n, g)
        assert rpr == cpr * 1024 and rpr * g >= n
        assert rpr * g - n < 1024 * g            # never more than one chunk per rank of slack
    assert icp_amd.rows_per_rank(1000000, 8) == 125952


def test_read_csv_matches_reference_behaviour(tmp_path):
    g = golden("c1_30pts")
    def w(name, text):
        p = tmp_path / name
        p.write_text(text)
        return p
    body = "".join("%.17g,%.17g,%.17g\n" % tuple(r) for r in g["P"])
    a, h = icp_amd.read_csv(w("h.csv", "Points_0,Points_1,Points_2\n" + body))
    assert np.array_equal(a, g["P"]) and h == "Points_0,Points_1,Points_2\n"
    # reference quirk (csv.cpp:113-114,128-129,147-148): without a header line the second line is taken for the header
    # text and the row count comes out one short -- the last row is lost (a one-line file keeps its row)
    a, h = icp_amd.read_csv(w("nh.csv", body))
    assert np.array_equal(a, g["P"][:-1]) and h == "%.17g,%.17g,%.17g\n" % tuple(g["P"][1])
    a, h = icp_amd.read_csv(w("one.csv", "1,2,3\n"))
    assert np.array_equal(a, [[1, 2, 3]]) and h == ""
    a, _ = icp_amd.read_csv(w("ws.csv", "x,y,z\n1.5, 2.5 ,3abc\n4,5,6\n"))       # strtod semantics: prefix parse, blanks
    assert np.array_equal(a, [[1.5, 2.5, 3.0], [4, 5, 6]])
    a, _ = icp_amd.read_csv(w("five.csv", "a,b,c,d,e\n1,2,3,4,5\n"))
    assert a.shape == (1, 5)
    for bad in ("x,y,z\n1,2,3\n\n", "1,2,3\nfoo,2,3\n", "1,2,3\n4,5\n", "", "x,y,z\n", "1,b,3\n"):
        with pytest.raises(icp_amd.IcpAmdError):
            icp_amd.read_csv(w("bad.csv", bad))
    with pytest.raises(icp_amd.IcpAmdError) as ei:
        icp_amd.read_csv(tmp_path / "missing.csv")
    assert "File could not be opened" in str(ei.value)
    if os.path.isdir(REF_DATA):   # the reference's own files, read in place (build container only)
        a, _ = icp_amd.read_csv(REF_DATA + "/30points_notebook_P.txt")
        assert np.array_equal(a, g["P"])
        a, _ = icp_amd.read_csv(REF_DATA + "/icp/cow_tr2.txt")
        assert a.shape == (2903, 5)


@pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "ref_trace")), reason="needs oracle/_ref (build container)")
def test_csv_readers_against_the_reference_reader_differential():
    """tests/fuzz/fuzz_csv.py, short: random small files through the reference's own readCSV (compiled unmodified into
    oracle/_ref/ref_trace) and through the three CSV entry points.  Same outcome, shape and bits on every file the
    library promises to read like the reference; an error (never data) on the files it documents as rejected."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "fuzz"))
    import fuzz_csv
    assert fuzz_csv.run(150, seed=11, verbose=False) == []


def _same_outcome(path):
    """serial and threaded parsers: same array + header, or the same error text"""
    outs = []
    for fn in (lambda: icp_amd.read_csv(path), lambda: icp_amd.read_csv_mt(path, 1), lambda: icp_amd.read_csv_mt(path, 5),
               lambda: icp_amd.read_csv_mt(path, 0), lambda: icp_amd.read_cloud(path)):
        try:
            a, h = fn()
            outs.append(("ok", a.shape, a.tobytes(), h))
        except icp_amd.IcpAmdError as e:
            outs.append(("err", str(e)))
    assert all(o == outs[0] for o in outs[1:]), (path, [o[:2] for o in outs])
    return outs[0]


def test_threaded_csv_parser_is_the_serial_one(tmp_path):
    """identical values, shapes, headers and errors, file by file."""
    rng = np.random.RandomState(4)
    def w(name, text):
        p = tmp_path / name
        p.write_bytes(text.encode())
        return p
    A = rng.uniform(-1e3, 1e3, (40000, 3))
    body = "".join("%.17g,%.17g,%.17g\n" % tuple(r) for r in A)          # ~2.3 MB: several 64 KB pieces per thread
    kind, shape, data, h = _same_outcome(w("big.csv", "Points_0,Points_1,Points_2\n" + body))
    assert kind == "ok" and shape == (40000, 3) and data == A.tobytes() and h == "Points_0,Points_1,Points_2\n"
    # header-less files: one row short, like the reference (see test_read_csv_matches_reference_behaviour)
    assert _same_outcome(w("nonl.csv", body[:-1]))[1] == (39999, 3)        # no newline at the end of the file
    assert _same_outcome(w("crlf.csv", body.replace("\n", "\r\n")))[1] == (39999, 3)
    assert _same_outcome(w("hcrlf.csv", "x,y,z\r\n" + body.replace("\n", "\r\n")))[1] == (40000, 3)
    lines = body.split("\n")[:-1]
    for k, (name, mutate) in enumerate((
            ("ragged_late", lambda L: L[:30000] + ["1,2"] + L[30000:]),
            ("ragged_early", lambda L: L[:3] + ["1,2,3,4"] + L[3:]),
            ("nan_text", lambda L: L[:20000] + ["1,abc,3"] + L[20000:]),
            ("blank_middle", lambda L: L[:25000] + [""] + L[25000:]),
            ("blank_end", lambda L: L + [""]),
            ("spaces", lambda L: L[:10] + [" 1.5 , 2.5 ,3abc", ",,4,,5,6,"] + L[10:]),
            ("two_errors", lambda L: L[:100] + ["1,2"] + L[100:35000] + ["x,y,z"] + L[35000:]))):
        out = _same_outcome(w("m%d.csv" % k, "\n".join(mutate(lines)) + "\n"))
        assert (out[0] == "ok") == (name == "spaces"), name
    for text in ("", "\n", "a,b,c\n", "1,2,3", "1,2,3\n", ",\n1,2\n", "h1,h2\n1,2\n3,4\n"):
        _same_outcome(w("tiny.csv", text))
    if os.path.isdir(REF_DATA):
        for name in ("30points_notebook_P.txt", "icp/cow_tr1.txt", "icp/cow_tr2.txt", "icp/horse_ref.txt"):
            assert _same_outcome(REF_DATA + "/" + name)[0] == "ok"


def test_binary_clouds_round_trip(tmp_path):
    rng = np.random.RandomState(6)
    A = rng.standard_normal((1234, 3)) * 10.0 ** rng.uniform(-200, 200, (1234, 3))
    for name in ("c.npy", "c.f64"):
        icp_amd.write_cloud_binary(tmp_path / name, A)
        B, _ = icp_amd.read_cloud(tmp_path / name)
        assert B.shape == A.shape and B.tobytes() == A.tobytes()
    assert np.array_equal(np.load(tmp_path / "c.npy"), A)                 # a real .npy, both ways
    np.save(tmp_path / "n.npy", A)
    assert np.array_equal(icp_amd.read_cloud(tmp_path / "n.npy")[0], A)
    np.save(tmp_path / "v.npy", A[:, 0].copy())
    assert icp_amd.read_cloud(tmp_path / "v.npy")[0].shape == (1234, 1)
    for bad, arr in (("f4.npy", A.astype(np.float32)), ("fo.npy", np.asfortranarray(A)), ("3d.npy", A.reshape(617, 2, 3))):
        np.save(tmp_path / bad, arr)
        with pytest.raises(icp_amd.IcpAmdError):
            icp_amd.read_cloud(tmp_path / bad)
    (tmp_path / "odd.f64").write_bytes(b"\0" * 25)
    with pytest.raises(icp_amd.IcpAmdError):
        icp_amd.read_cloud(tmp_path / "odd.f64")
    with pytest.raises(icp_amd.IcpAmdError):
        icp_amd.write_cloud_binary(tmp_path / "c.txt", A)


def test_write_cloud_and_report_format(tmp_path):
    g = golden("c1_30pts")
    out = tmp_path / "point_cloud_result"
    icp_amd.write_cloud(out, g["P_out"])
    lines = out.read_text().split("\n")
    assert lines[0] == " dim0: 30 dim1: 3 array: "
    assert lines[1] == " ".join("%g" % v for v in g["P_out"][0]) + " "
    d = icp_amd.euclidian_distance(g["Q"], g["P_out"])
    assert d == orc.euclidian_distance(g["Q"], g["P_out"])
    assert "%g" % d == "3.48782e-14"          # the value CPUICP prints for this pair
    with pytest.raises(icp_amd.IcpAmdError):
        icp_amd.euclidian_distance(g["Q"], g["P_out"][:5])


@pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "CPUICP")), reason="needs oracle/_ref (build container)")
def test_writer_and_report_against_the_reference_executable():
    """tests/fuzz/fuzz_writer.py, short: `point_cloud_result` and the "Squared actual mean diff" line of the reference's own
    CPUICP (unmodified main.cpp) are reproduced byte for byte from the doubles it held, across scales 1e-8 .. 1e8."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "fuzz"))
    import fuzz_writer
    compared, bad = fuzz_writer.run(40, seed=5, verbose=False)
    assert compared >= 30 and bad == []


def test_cli_argument_errors():
    exe = os.path.join(ROOT, "bin", "GPUICP")
    r = subprocess.run([exe, "a", "b"], capture_output=True, text=True)
    assert r.returncode != 0
    assert "Usage: ./GPUICP file1 file2 nbiters method={-loop|-batch|-shared|-shared-loop}" in r.stderr
    r = subprocess.run([exe, "/nonexistent", "/nonexistent", "3", "-loop"], capture_output=True, text=True)
    assert r.returncode != 0 and "File could not be opened !" in r.stderr


def test_cpp_adaptor_compiles_against_reference_types(tmp_path):
    """include/icp_amd.hpp keeps the reference's C++ signatures (include/gpu/icp.cuh:38-39): instantiate it with the
    reference's own CPUMatrix (build container only; header-only syntax check, nothing is linked or run)."""
    ref_inc = "/root/reference/include"
    if not os.path.isdir(ref_inc):
        pytest.skip("reference not mounted")
    src = tmp_path / "adaptor.cpp"
    src.write_text(
        '#include "libalg/CPUMatrix.hpp"\n#include "icp_amd.hpp"\n'
        'CPUMatrix f(CPUMatrix &P, CPUMatrix &Q, std::string &m) { return icp_amd::icp_gpu_optimized(P, Q, 3u, m); }\n'
        'CPUMatrix g(CPUMatrix &P, CPUMatrix &Q) { return icp_amd::icp_gpu(P, Q, 3u); }\n')
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-I", ref_inc, "-I", os.path.join(ROOT, "include"), str(src)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_tie_band_covers_every_key_preimage():
    """The search kernels treat every d^2 within 2^-48 (relative, + 1e-300) of the best as a possible key tie
    (exact.h tie_band).  The key is sqrt(sqrt(d^2)), correctly rounded twice: the doubles that share one key form a run
    of at most 9, i.e. less than half the band."""
    rng = np.random.RandomState(0)
    d = np.concatenate([rng.uniform(1, 16, 200000), 10.0 ** rng.uniform(-300, 300, 200000), rng.uniform(0, 1e-305, 1000)])
    k0 = np.sqrt(np.sqrt(d))
    hi = d.copy()
    alive = np.ones(len(d), bool)
    top = d.copy()
    for _ in range(24):
        hi = np.nextafter(hi, np.inf)
        alive &= np.sqrt(np.sqrt(hi)) == k0
        top = np.where(alive, hi, top)
    assert not alive.any()                                  # every run ended within 24 steps
    assert np.all(top <= d * (1 + 2.0 ** -48) + 1e-300)     # and inside the band
