#!/usr/bin/env python3
"""Generate the committed golden vectors under tests/golden/ from the UNMODIFIED reference.

Runs oracle/_ref/ref_trace (oracle/ref_trace.cpp linked against the reference's own CPU sources, built by
oracle/Makefile where they lie under /root/reference) on reference data files and on synthetic /
adversarial clouds, and packs inputs + per-iteration outputs into small .npz files.  Only data is stored
(inputs and expected outputs); no reference source travels.  Re-run only in the build container:

    make -C oracle ref && python tests/golden/make_golden.py [case ...]
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from icp_amd import synth  # noqa: E402

REF = "/root/reference/data"
TRACE = os.path.join(ROOT, "oracle", "_ref", "ref_trace")


def load_csv(path):
    rows = []
    with open(path) as f:
        for k, line in enumerate(f):
            tok = line.strip().split(",")
            try:
                rows.append([float(x) for x in tok])
            except ValueError:
                if k != 0:
                    raise
    return np.array(rows, dtype=np.float64)


def write_csv(path, A):
    with open(path, "w") as f:
        f.write("Points_0,Points_1,Points_2\n")
        for r in A:
            f.write("%.17g,%.17g,%.17g\n" % tuple(r))


def read_trace(path):
    raw = open(path, "rb").read()
    assert raw[:8] == b"ICPTRACE"
    n, m, iters, flags = np.frombuffer(raw, dtype=np.uint64, count=4, offset=8).astype(int)
    off = 40
    def take(count, dtype=np.float64):
        nonlocal off
        a = np.frombuffer(raw, dtype=dtype, count=count, offset=off)
        off += a.nbytes
        return a.copy()
    out = {"Qc": take(3)}
    keys = ("P_in", "Pc", "idx", "cov", "S", "U", "VT", "R", "t")
    acc = {k: [] for k in keys}
    for _ in range(iters):
        if flags & 1:
            acc["P_in"].append(take(n * 3).reshape(n, 3))
        acc["Pc"].append(take(3))
        acc["idx"].append(take(n, np.int32))
        acc["cov"].append(take(9).reshape(3, 3))
        acc["S"].append(take(3))
        acc["U"].append(take(9).reshape(3, 3))
        acc["VT"].append(take(9).reshape(3, 3))
        acc["R"].append(take(9).reshape(3, 3))
        acc["t"].append(take(3))
    out["P_out"] = take(n * 3).reshape(n, 3)
    assert off == len(raw)
    for k in keys:
        if acc[k]:
            out[k] = np.stack(acc[k])
    return out


def run_icp_case(name, P, Q, iters, keep_p=None, pfile=None, qfile=None, store_inputs=True):
    with tempfile.TemporaryDirectory() as td:
        if pfile is None:
            pfile, qfile = os.path.join(td, "P.csv"), os.path.join(td, "Q.csv")
            write_csv(pfile, P)
            write_csv(qfile, Q)
        out = os.path.join(td, "trace.bin")
        subprocess.check_call([TRACE, pfile, qfile, str(iters), out, "1"])
        tr = read_trace(out)
    # the reference loop starts from the parsed input: pin our parse / csv round-trip against it
    assert np.array_equal(tr["P_in"][0], P), name
    keep = list(range(iters)) if keep_p is None else [k for k in keep_p if k < iters]
    tr["P_in_iters"] = np.array(keep, dtype=np.int32)
    # iteration 0 starts from P itself (asserted above): not stored twice; conftest.golden() re-inserts it
    tr["P_in"] = tr["P_in"][[k for k in keep if k > 0]]
    if not store_inputs:
        # the inputs are those of the one-iteration fixture of the same pair (<name without _x20>.npz): not stored twice
        tr["inputs_from"] = np.array(name.rsplit("_x", 1)[0])
        np.savez_compressed(os.path.join(HERE, name + ".npz"), iters=np.int32(iters), **tr)
    else:
        np.savez_compressed(os.path.join(HERE, name + ".npz"), P=P, Q=Q, iters=np.int32(iters), **tr)
    print("%-14s N=%d M=%d iters=%d  -> %s.npz" % (name, len(P), len(Q), iters, name))


CKPT = os.environ.get("GOLDEN_CKPT", "/tmp/golden_ckpt")


def run_icp_case_resumable(name, P, Q, iters, keep_p, inputs_from, store_p=False):
    """run_icp_case for the long runs (half an hour of reference time): one ref_trace process per iteration, each started
    from the cloud the previous one wrote (a %.17g CSV round trip is exact, and the reference re-centres the same Q every
    time, so the chain equals one long run bit for bit), with every iteration's outputs kept under $GOLDEN_CKPT so that an
    interrupted generation picks up where it stopped."""
    os.makedirs(CKPT, exist_ok=True)
    keys = ("Pc", "idx", "cov", "S", "U", "VT", "R", "t")
    acc = {k: [] for k in keys}
    p_in = []
    cur, qc = P, None
    with tempfile.TemporaryDirectory() as td:
        qfile = os.path.join(td, "Q.csv")
        write_csv(qfile, Q)
        for it in range(iters):
            ck = os.path.join(CKPT, "%s_it%02d.npz" % (name, it))
            if os.path.exists(ck):
                with np.load(ck) as z:
                    tr = {k: z[k] for k in z.files}
            else:
                pfile, out = os.path.join(td, "P.csv"), os.path.join(td, "trace.bin")
                write_csv(pfile, cur)
                subprocess.check_call([TRACE, pfile, qfile, "1", out, "1"])
                tr = read_trace(out)
                assert np.array_equal(tr["P_in"][0], cur), (name, it)
                tr = {k: (tr[k][0] if k in keys else tr[k]) for k in list(keys) + ["Qc", "P_out"]}
                np.savez(ck + ".tmp.npz", **tr)
                os.replace(ck + ".tmp.npz", ck)
                print("%-14s iteration %d done" % (name, it), flush=True)
            if it in keep_p and it > 0:
                p_in.append(cur)
            for k in keys:
                acc[k].append(tr[k])
            qc, cur = tr["Qc"], tr["P_out"]
    res = {k: np.stack(acc[k]) for k in keys}
    res.update(Qc=qc, P_out=cur, P_in=np.stack(p_in), P_in_iters=np.array([k for k in keep_p if k < iters], dtype=np.int32),
               inputs_from=np.array(inputs_from))
    if store_p:   # the source cloud is this fixture's own; only the target comes from `inputs_from`
        res["P"] = P
    np.savez_compressed(os.path.join(HERE, name + ".npz"), iters=np.int32(iters), **res)
    print("%-14s N=%d M=%d iters=%d  -> %s.npz" % (name, len(P), len(Q), iters, name))


def ties_clouds():
    """Adversarial correspondence-only case (SURVEY.md 7.3 hard parts 1-2): inputs are used as-is
    (already 'centred'), so exact and few-ulp ties survive."""
    rng = np.random.RandomState(7)
    P, Q = [], []
    # (a) known answer: d^2 = {1+2^-52.., 1}: both fourth roots round to 1 -> the FIRST (farther) wins
    P.append([0.0, 0.0, 0.0])
    Q += [[np.nextafter(1.0, 2.0), 0.0, 0.0], [1.0, 0.0, 0.0]]
    # (b) half-integer lattice: every p is exactly equidistant from 8 lattice points
    g = np.arange(6, dtype=np.float64) * 0.5 + 3.0
    lat = np.array([[x, y, z] for x in g for y in g for z in g])
    Q += lat[rng.permutation(len(lat))].tolist()
    P += (lat[:125] + 0.25).tolist()
    # (c) exact duplicates in Q at scattered positions, and p's sitting exactly on them
    off = np.array([0.0, 0.0, -50.0])   # keep this block away from the known-answer pair at the origin
    dup = rng.uniform(-2, 2, size=(40, 3)) + off
    Q += dup.tolist() + (rng.uniform(-2, 2, size=(100, 3)) + off).tolist() + dup[::-1].tolist()
    P += dup.tolist()
    # (d) candidates whose d^2 differ by 1..8 ulp, the slightly farther one listed first
    for k in range(1, 9):
        for _ in range(12):
            p = rng.uniform(-1, 1, 3)
            d = rng.uniform(0.5, 2.0)
            q_near = p + np.array([d, 0, 0])
            q_far = q_near.copy()
            for _ in range(k):
                q_far[0] = np.nextafter(q_far[0], 10.0)
            Q += [(q_far + 20.0).tolist(), (q_near + 20.0).tolist()]
            P.append((p + 20.0).tolist())
    # (e) mirrored pairs about p (exact real-arithmetic ties along each axis)
    for _ in range(60):
        p = np.round(rng.uniform(-4, 4, 3) * 64) / 64 - 30.0
        o = np.round(rng.uniform(0.1, 1, 3) * 64) / 64
        Q += [(p + o).tolist(), (p - o).tolist(), (p + o * [1, -1, 1]).tolist(), (p - o * [1, -1, 1]).tolist()]
        P.append(p.tolist())
    return np.array(P, dtype=np.float64), np.array(Q, dtype=np.float64)


def run_corr_case(name, Pz, Qz):
    with tempfile.TemporaryDirectory() as td:
        pf, qf, out = os.path.join(td, "P.csv"), os.path.join(td, "Q.csv"), os.path.join(td, "idx.bin")
        write_csv(pf, Pz)
        write_csv(qf, Qz)
        subprocess.check_call([TRACE, "--corr", pf, qf, out])
        idx = np.fromfile(out, dtype=np.int32)
    assert len(idx) == len(Pz)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), Pz=Pz, Qz=Qz, idx=idx)
    print("%-14s N=%d M=%d (correspondences only) -> %s.npz" % (name, len(Pz), len(Qz), name))


def lattice_pair():
    """Whole-loop case with structured exact ties: Q a 10x10x10 integer lattice (duplicated corner
    points), P the same lattice rotated slightly and offset by half a cell."""
    g = np.arange(10, dtype=np.float64)
    Q = np.array([[x, y, z] for x in g for y in g for z in g])
    Q = np.concatenate([Q, Q[:50]])
    P = Q[:1000] @ synth.rotation(0.02).T + np.array([0.5, 0.5, 0.5])
    P[:200] = Q[:200] + 0.5   # exact half-cell offsets: 8-way ties
    return np.ascontiguousarray(P), np.ascontiguousarray(Q)


CASES = {}


def case(fn):
    CASES[fn.__name__] = fn
    return fn


@case
def c1_30pts():
    pf, qf = REF + "/30points_notebook_P.txt", REF + "/30points_notebook_Q.txt"
    run_icp_case("c1_30pts", load_csv(pf), load_csv(qf), 5, pfile=pf, qfile=qf)


@case
def cow_tr1():
    pf, qf = REF + "/icp/cow_tr1.txt", REF + "/icp/cow_ref.txt"
    run_icp_case("cow_tr1", load_csv(pf), load_csv(qf), 20, keep_p=[0, 1, 2, 10, 19], pfile=pf, qfile=qf)


# (cow_tr2.txt has 5 columns: the reference aborts on it -- "element_wise_reduce requires arrays to be
#  broadcastable" -- so it is an error-path case, not a golden one.)


@case
def horse_tr1():
    pf, qf = REF + "/icp/horse_tr1.txt", REF + "/icp/horse_ref.txt"
    run_icp_case("horse_tr1", load_csv(pf), load_csv(qf), 1, pfile=pf, qfile=qf)


@case
def bun045_000():
    pf, qf = REF + "/icp/bun045.txt", REF + "/icp/bun000.txt"
    run_icp_case("bun045_000", load_csv(pf), load_csv(qf), 1, pfile=pf, qfile=qf)


# BASELINE configs[1] at its full length: 20 iterations of the reference loop on the two largest sample pairs
# (about half an hour of reference CPU time each; the one-iteration fixtures above stay for the quick tests).
@case
def horse_tr1_x20():
    pf, qf = REF + "/icp/horse_tr1.txt", REF + "/icp/horse_ref.txt"
    run_icp_case_resumable("horse_tr1_x20", load_csv(pf), load_csv(qf), 20, [0, 1, 10, 19], "horse_tr1")


@case
def bun045_000_x20():
    pf, qf = REF + "/icp/bun045.txt", REF + "/icp/bun000.txt"
    run_icp_case_resumable("bun045_000_x20", load_csv(pf), load_csv(qf), 20, [0, 1, 10, 19], "bun045_000")


# The rest of data/icp (BASELINE configs[1] is "data/icp sample clouds"): the second horse pose for the full 20 iterations,
# and the three remaining bunny scans against bun000 (N != M: 40251 / 31701 / 35336 source rows against 40256 targets) for one
# iteration of the reference loop each.  Targets are those of horse_tr1 / bun045_000 and are not stored again.
@case
def horse_tr2_x20():
    pf, qf = REF + "/icp/horse_tr2.txt", REF + "/icp/horse_ref.txt"
    run_icp_case_resumable("horse_tr2_x20", load_csv(pf), load_csv(qf), 20, [0, 10], "horse_tr1", store_p=True)


def bun_case(scan):
    pf, qf = REF + "/icp/%s.txt" % scan, REF + "/icp/bun000.txt"
    P, Q = load_csv(pf), load_csv(qf)
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "trace.bin")
        subprocess.check_call([TRACE, pf, qf, "1", out, "1"])
        tr = read_trace(out)
    assert np.array_equal(tr["P_in"][0], P), scan
    name = scan + "_000"
    keep = {k: tr[k] for k in ("Qc", "Pc", "idx", "cov", "S", "R", "t")}
    # P_out is left out (1 MB of incompressible doubles per scan): the transform apply is pinned bit for bit elsewhere
    np.savez_compressed(os.path.join(HERE, name + ".npz"), P=P, inputs_from=np.array("bun045_000"), iters=np.int32(1), **keep)
    print("%-14s N=%d M=%d iters=1 -> %s.npz" % (name, len(P), len(Q), name))


@case
def bun180_000():
    bun_case("bun180")


@case
def bun270_000():
    bun_case("bun270")


@case
def bun315_000():
    bun_case("bun315")


@case
def syn256():
    P, Q = synth.make_pair(256, 256, seed=101)
    run_icp_case("syn256", P, Q, 5)


@case
def syn1k():
    P, Q = synth.make_pair(1000, 1000, seed=102)
    run_icp_case("syn1k", P, Q, 5, keep_p=[0, 1, 4])


@case
def syn3k_4k():
    P, Q = synth.make_pair(3000, 4099, seed=103)
    run_icp_case("syn3k_4k", P, Q, 3, keep_p=[0, 2])


@case
def syn20k():
    P, Q = synth.make_pair(20000, 20000, seed=104)
    run_icp_case("syn20k", P, Q, 2, keep_p=[0, 1])


@case
def lattice():
    P, Q = lattice_pair()
    run_icp_case("lattice", P, Q, 3)


@case
def ties_corr():
    Pz, Qz = ties_clouds()
    run_corr_case("ties_corr", Pz, Qz)


def run_rows_case(name, n, iters_at, n_rows, threads):
    """Sampled rows of a full-size synthetic pair (the bench's own clouds, synth.make_pair(n, n)) through the REFERENCE's
    get_correspondence_indices (src/cpu/icp.cpp:72-93) at the iterations `iters_at` (0-based).  The reference cannot walk
    the whole loop at this size (10 h per iteration at 1M x 1M), so the running cloud P_k comes from the oracle port
    (oracle/icp_oracle.c, pinned bit-exact to the reference by the other fixtures) and the reference answers for
    `n_rows` sampled rows against ALL n targets.  Stored: the sampled rows of P_k, Pc_k, the reference's indices, and the
    previous iteration's indices of those rows (the seeds a seeded search launch starts from).  P and Q themselves are not
    stored: they are synth.make_pair(n, n)."""
    from oracle import pyoracle as orc
    P, Q = synth.make_pair(n, n)
    last = max(iters_at)
    # the oracle port's trajectory, one iteration per call with a checkpoint each (an iteration at 1M x 1M is minutes):
    # the chain of one-iteration calls is the loop itself (each starts from the previous call's P_out, Q is re-centred alike)
    os.makedirs(CKPT, exist_ok=True)
    tr = {"P_in": [], "idx": []}
    cur = P
    for it in range(last):
        ck = os.path.join(CKPT, "%s_port_it%02d.npz" % (name, it))
        if os.path.exists(ck):
            with np.load(ck) as z:
                one = {k: z[k] for k in z.files}
        else:
            r1 = orc.icp(cur, Q, 1, trace=True, nthreads=threads)
            one = {"idx": r1["idx"][0], "P_out": r1["P_out"]}
            np.savez(ck + ".tmp.npz", **one)
            os.replace(ck + ".tmp.npz", ck)
            print("%-14s oracle-port iteration %d done" % (name, it), flush=True)
        tr["P_in"].append(cur)
        tr["idx"].append(one["idx"])
        cur = one["P_out"]
    tr["P_out"] = cur
    Qc = orc.mean0(Q)
    Qz = orc.sub_row(Q, Qc)
    rows = np.sort(np.random.RandomState(n % 1000003).choice(n, n_rows, replace=False)).astype(np.int64)
    out = {"n": np.int64(n), "rows": rows, "iters_at": np.array(iters_at, dtype=np.int32), "Qc": Qc}
    with tempfile.TemporaryDirectory() as td:
        qf = os.path.join(td, "Q.csv")
        write_csv(qf, Qz)
        for k in iters_at:
            Pk = P if k == 0 else (tr["P_in"][k] if k < last else tr["P_out"])
            Pc = orc.mean0(Pk)
            Pz = orc.sub_row(Pk[rows], Pc)
            pf, of = os.path.join(td, "P%d.csv" % k), os.path.join(td, "idx%d.bin" % k)
            write_csv(pf, Pz)
            subprocess.check_call([TRACE, "--corr", pf, qf, of])
            idx = np.fromfile(of, dtype=np.int32)
            assert len(idx) == n_rows
            port = orc.correspondences(Pz, Qz, nthreads=threads)
            assert np.array_equal(idx, port), "oracle port differs from the reference at full size"
            out["P_rows_%d" % k] = Pk[rows].copy()
            out["Pc_%d" % k] = Pc
            out["idx_%d" % k] = idx
            if k > 0:
                out["seed_%d" % k] = tr["idx"][k - 1][rows].copy()
            print("%-14s iteration %d: %d rows x %d targets through the reference" % (name, k, n_rows, n), flush=True)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)


# BASELINE configs[2] (100k x 100k, 30 iterations): iterations 1, 15 and 30
@case
def syn100k_rows():
    run_rows_case("syn100k_rows", 100000, [0, 14, 29], 2000, int(os.environ.get("GOLDEN_THREADS", "8")))


# BASELINE configs[3] (1M x 1M): iteration 1 (un-seeded) and iteration 3 (seeded)
@case
def syn1m_rows():
    run_rows_case("syn1m_rows", 1000000, [0, 2], 400, int(os.environ.get("GOLDEN_THREADS", "8")))


if __name__ == "__main__":
    names = sys.argv[1:] or list(CASES)
    for nm in names:
        CASES[nm]()
