#!/usr/bin/env python3
"""bench.py -- the reference's headline metric on MI355X: ICP iterations/s and distance evaluations/s at
N = M = 1M (BASELINE.json configs[3]; source rows sharded over the GPUs, target replicated), fp64.

    python bench.py --gpus 1 --steps 10 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step is one full ICP iteration (centroid, brute-force correspondence search, cross-covariance, 3x3 solve,
transform apply) on device-resident clouds.  Rank 0 prints ONE JSON line.  The timed region is the iteration
loop only, bracketed by barrier + device synchronisation on both sides, max over ranks.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP64_VALU_NOFMA = 39.3e12   # 256 CU x 4 SIMD x 16 fp64 lanes/clk x 2.4 GHz, one flop per op
PEAK_FP32_VALU = 157.3e12        # MI355X fp32 vector peak (MI355X_MICROARCH.md; packed FMA counted as 2 flops per lane)
FLOPS_PER_PAIR = 8               # 3 sub, 3 mul, 2 add: the oracle's un-fused distance
PEAK_F16_MFMA = 2500e12          # MI355X dense f16 / bf16 matrix peak (MI355X_MICROARCH.md: ~2.5 PF dense)
CUS, CLOCK_HZ = 256, 2.4e9


def blob_hash(path):
    """What `git hash-object` gives the file (no git needed on the GPU box)."""
    import hashlib
    data = open(path, "rb").read()
    return hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()


def measured_traffic(kernel, n, m, world):
    """(bytes per launch, where from, blob hash of the kernel source they were measured on) from the committed PMC passes
    (icp_amd/measured_traffic.json; written by tools/update_traffic.py), or (None, why, None): bench.py cannot collect
    counters itself (rocprofv3 runs them as separate passes), so the figure is only reported when it was taken on this
    workload AND on the kernel source this tree holds."""
    try:
        rec = json.load(open(os.path.join(ROOT, "icp_amd", "measured_traffic.json"))).get(kernel)
    except (OSError, ValueError):
        return None, "no measured_traffic.json", None
    if not rec or (rec["n"], rec["m"] if m is not None else None, rec["n_gpus"]) != (n, m, world):
        return None, "no PMC pass on this workload", None
    src = rec.get("kernel_source")
    try:
        now = blob_hash(os.path.join(ROOT, src)) if src else None
    except OSError:
        now = None
    if not now or now != rec.get("kernel_source_blob"):
        return None, "PMC pass predates the current %s (blob %s, measured on %s)" % (src, now, rec.get("kernel_source_blob")), None
    return rec["traffic_bytes"], rec.get("source", ""), now


def roofline(variant, n_local, M, avg_ms, launches):
    """Roofline object of the search kernel.  `achieved` is ALGORITHMIC: 8 flops per (p,q) pair x the
    pairs of one launch / the kernel's hipEvent-timed average launch duration, whatever instruction mix the variant's
    fast path really issues; `peak` is the VALU peak of the arithmetic type that fast path runs in."""
    ach = FLOPS_PER_PAIR * float(n_local) * float(M) / (avg_ms * 1e-3) if avg_ms > 0 else 0.0
    if variant == "f32" and os.environ.get("ICP_AMD_MFMA", "16") == "16":
        # the default F32 filter: one v_mfma_f32_32x32x16_f16 per 32 targets x 32 source rows on operands in two fp16 pieces
        pairs = float(n_local) * float(M)
        issued = 32.0 * pairs / (avg_ms * 1e-3) if avg_ms > 0 else 0.0        # K = 16: 32 matrix flops issued per pair
        valu_floor_ms = pairs * 0.5 / (CUS * 4 * 16 * CLOCK_HZ) * 1e3           # half a v_min3_f32 per pair at one lane-op per lane-clock
        return {"bound": "mfma", "kernel": "search_mfma16_kernel", "achieved": ach / 1e12, "peak": PEAK_F16_MFMA / 1e12,
                "unit": "TFLOP/s", "frac": ach / PEAK_F16_MFMA, "flops_per_pair": FLOPS_PER_PAIR, "pairs_per_launch": pairs,
                "avg_launch_ms": avg_ms, "launches": launches, "traffic": None, "variant": variant,
                "mfma_flops_issued_per_pair": 32, "mfma_issued_tflops": issued / 1e12, "mfma_pipe_frac": issued / PEAK_F16_MFMA,
                "valu_floor_ms": valu_floor_ms, "frac_of_valu_floor": valu_floor_ms / avg_ms if avg_ms > 0 else 0.0,
                "instruction_mix": "fast path = fp16 MFMA pre-filter (operands split into hi + lo fp16 pieces, 11 of the 16 K slots "
                                   "non-zero) + 0.5 v_min3_f32 per pair on the VALU, co-executing; every candidate re-evaluated in "
                                   "un-fused fp64 (indices bit-identical to the fp64 variants)",
                "note": "`achieved` counts the ALGORITHMIC 8 flops per pair against the dense f16 MFMA peak (2.5 PFLOP/s); the "
                        "matrix pipe itself is `mfma_pipe_frac` busy at the nominal 2.4 GHz (32 issued flops per pair).  The kernel "
                        "is POWER bound: its MFMA stream alone runs at 1.63 GHz on random operands (20.7 ms for this launch) "
                        "against 2.39 GHz on zeros (13.8 ms), profiles/r2_ubench_mfma.txt; `valu_floor_ms` is the time of its half "
                        "v_min3_f32 per pair alone"}
    f32 = variant == "f32"
    peak = PEAK_FP32_VALU if f32 else PEAK_FP64_VALU_NOFMA
    mix = {"f32": "fast path = fp32 pre-filter, 3 v_fma_f32 (or 1.5 v_pk_fma_f32) + 0.5 v_min3_f32 per pair; every candidate "
                  "re-evaluated in un-fused fp64 (indices bit-identical to the fp64 variants)",
           "fma": "fast path = 6.58 VALU instructions per pair (3 add, 1 mul, 2 fma in fp64 + half a v_min3_u32 on the high words); candidates re-evaluated un-fused",
           "exact": "fast path = 8.58 VALU instructions per pair (the oracle's 8 un-fused fp64 ops + half a v_min3_u32 on the high words of two d^2)"}.get(variant, variant)
    return {"bound": "valu_fp32" if f32 else "valu_fp64", "kernel": "search_f32_kernel" if f32 else "search_kernel",
            "achieved": ach / 1e12, "peak": peak / 1e12, "unit": "TFLOP/s", "frac": ach / peak,
            "flops_per_pair": FLOPS_PER_PAIR, "pairs_per_launch": float(n_local) * float(M), "avg_launch_ms": avg_ms,
            "launches": launches, "traffic": None, "variant": variant, "instruction_mix": mix,
            "note": ("fp32 vector peak 157.3 TFLOP/s" if f32 else
                     "un-fused fp64 VALU peak 39.3 TFLOP/s = one flop per op (FMA-counted datasheet peak: 78.6); the FMA "
                     "variant issues 6.58 instructions for the 8 algorithmic flops, so its frac can pass 1")}


# what the box's memory system delivers for plain streams at the grid iteration's sizes (tools/ubench_stream.hip at 1M rows,
# profiles/r3_ubench_stream.txt): a copy (32 B in, 32 B out per row) and two read streams + one write stream
MEASURED_COPY_TBPS, MEASURED_2R1W_TBPS = 5.69, 6.83
GRID_SEARCH_BYTES_PER_ROW = 100      # settled search + covariance: record halves A, B in (64), B out (32), index out (4)
GRID_ITERATION_BYTES_PER_ROW = 164   # + transform apply: half A in, half A out (64); solve and centroid move nothing per row


def grid_roofline(n_local, kernel_ms, step_s, world):
    ach = GRID_SEARCH_BYTES_PER_ROW * n_local / (kernel_ms * 1e-3) if kernel_ms > 0 else 0.0
    it = GRID_ITERATION_BYTES_PER_ROW * n_local / step_s if step_s > 0 else 0.0
    tb, src, blob = measured_traffic("search_grid_settled_kernel", n_local, None, world)
    return {"bound": "hbm", "kernel": "search_grid_settled_kernel (search + covariance sums in one launch)", "achieved": ach / 1e9,
            "peak": 8000.0, "unit": "GB/s", "frac": ach / 8e12, "algorithmic_bytes_per_row": GRID_SEARCH_BYTES_PER_ROW,
            "avg_launch_ms": kernel_ms, "traffic": tb, "traffic_source_commit": blob, "traffic_note": src,
            "frac_of_measured_2r1w_rate": ach / (MEASURED_2R1W_TBPS * 1e12),
            "iteration_bytes_per_row": GRID_ITERATION_BYTES_PER_ROW, "iteration_frac": it / 8e12,
            "iteration_frac_of_measured_copy_rate": it / (MEASURED_COPY_TBPS * 1e12),
            "measured_stream_rates_tbps": {"copy": MEASURED_COPY_TBPS, "two_reads_one_write": MEASURED_2R1W_TBPS,
                                           "source": "tools/ubench_stream.hip at 1M rows, profiles/r3_ubench_stream.txt"},
            "note": "`iteration_*`: the two launches of a settled iteration that move rows (search + centroid + covariance 100 B per "
                    "row, apply 64) against the whole step time, which also holds the one-workgroup solve (10 us)"}


def fp64_variants(P, Q, device, steps=3):
    """The same workload through the two fp64-only search variants (N = 1 only), for the fp64-VALU roofline the
    north star names.  Results are bit-identical to the default path (asserted by tests/test_gpu_parity.py)."""
    import icp_amd
    out = {}
    for var in ("fma", "exact"):
        e = icp_amd.Engine(device=device, search=var, kernel_timing=True)
        e.set_target(Q)
        e.set_source(P)
        e.iterate(2)                      # first iteration is un-seeded
        e.sync()
        e.kernel_ms()
        ms = e.iterate_timed(steps) / steps
        k_ms, _, launches = e.kernel_ms()
        geo = e.search_geometry()
        e.close()
        out[var] = {"ms_per_step": ms, "value": float(len(P)) * len(Q) / (ms * 1e-3), "unit": "distance evals/s",
                    "search": geo, "roofline": roofline(var, len(P), len(Q), k_ms, launches)}
    return out


def parity_check(P, Q, device, rows=200000, iters=3):
    """In the bench run itself: the default brute-force search (reduced-precision pre-filter on the matrix pipe + exact fp64
    re-evaluation of every candidate) against the all-fp64 EXACT variant on a sub-sample of the same clouds.  Indices and
    every fp64 result must be identical, bit for bit."""
    import icp_amd
    n, m = min(rows, len(P)), min(rows, len(Q))
    got = {}
    for var in ("f32", "exact"):
        e = icp_amd.Engine(device=device, search=var)
        e.set_target(Q[:m])
        e.set_source(P[:n])
        e.iterate(iters)
        last = e.get_last()
        got[var] = (last["idx"].copy(), last["cov"].copy(), last["R"].copy(), last["t"].copy(), e.get_source())
        e.close()
    a, b = got["f32"], got["exact"]
    return {"workload": "first %d x %d points of the same clouds, %d iterations, default search against the all-fp64 EXACT variant"
                        % (n, m, iters),
            "indices_identical": bool(np.array_equal(a[0], b[0])),
            "cov_R_t_P_identical": bool(all(np.array_equal(x, y) for x, y in zip(a[1:], b[1:])))}


def cpu_baseline(P, Q, seconds_target=30.0):
    """The CPU oracle (oracle/icp_oracle.c, a port of the reference CPU path) timed on this host's cores on a
    bounded sample of the same workload: the correspondence search of the first `rows` source points against
    the whole target (the search is >99% of an iteration)."""
    from oracle import pyoracle as orc
    cores = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "0") or 0) or (os.cpu_count() or 1))
    Qz = Q - orc.mean0(Q)
    Pz = P - orc.mean0(P)
    m = len(Q)
    rows = max(cores, 16)
    t0 = time.perf_counter()
    orc.correspondences(Pz[:rows], Qz, nthreads=cores)
    dt = max(time.perf_counter() - t0, 1e-6)
    rows = int(min(len(P), max(rows, rows * seconds_target / dt)))
    t0 = time.perf_counter()
    orc.correspondences(Pz[:rows], Qz, nthreads=cores)
    dt = time.perf_counter() - t0
    return {"value": rows * m / dt, "unit": "distance evals/s", "cores": cores, "kind": "port",
            "sample": "correspondence search of the first %d source rows x %d targets, %d OpenMP threads, %.1f s"
                      % (rows, m, cores, dt),
            "iters_per_sec_extrapolated": (rows * m / dt) / (len(P) * float(m))}


def cpu_reference(P, Q, rows=4000, targets=20000):
    """The UNMODIFIED reference CPU code (oracle/_ref/ref_trace: the reference's own get_correspondence_indices,
    src/cpu/icp.cpp:72-93, compiled from its sources where they lie) timed on a bounded sample of the same clouds.
    Single-threaded, like CPUICP.  The CSV parse is taken out by subtracting a run with one source row."""
    import subprocess
    import tempfile
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_trace")
    if not os.path.exists(exe):
        return None
    from oracle import pyoracle as orc
    Pz = (P - orc.mean0(P))[:rows]
    Qz = (Q - orc.mean0(Q))[:targets]
    try:
        with tempfile.TemporaryDirectory() as d:
            def dump(name, A):
                path = os.path.join(d, name)
                np.savetxt(path, A, fmt="%.17g", delimiter=",", header="Points_0,Points_1,Points_2", comments="")
                return path
            pf, p1, qf, out = dump("p.csv", Pz), dump("p1.csv", Pz[:1]), dump("q.csv", Qz), os.path.join(d, "o.bin")
            times = []
            for src in (p1, pf):
                t0 = time.perf_counter()
                subprocess.run([exe, "--corr", src, qf, out], check=True, capture_output=True, timeout=300)
                times.append(time.perf_counter() - t0)
            idx = np.fromfile(out, dtype=np.int32)
        dt = max(times[1] - times[0], 1e-6)
        ok = bool(np.array_equal(idx[-len(Pz):], orc.correspondences(Pz, Qz, nthreads=4))) if len(idx) >= len(Pz) else None
        return {"value": (len(Pz) - 1) * float(len(Qz)) / dt, "unit": "distance evals/s", "cores": 1, "kind": "reference",
                "sample": "reference get_correspondence_indices on the first %d source rows x first %d targets of the same "
                          "clouds, %.1f s" % (len(Pz), len(Qz), dt), "indices_equal_oracle": ok,
                "note": "a sub-sample, NOT the %d x %d workload: at this rate the reference needs about %.1f h per full iteration "
                        "on one core" % (len(P), len(Q), len(P) * float(len(Q)) / ((len(Pz) - 1) * float(len(Qz)) / dt) / 3600.0)}
    except Exception as exc:   # the baseline is a report, never a reason for the bench to fail
        return {"error": repr(exc)[:200], "kind": "reference"}


def measure_batch(args, rank, world, local_rank, steps, warmup, with_cpu):
    """BASELINE configs[4]: args.batch independent registrations of n x n points, args.batch_iters iterations each, split
    evenly over the GPUs with no data-path collective.  One step = one launch of batch_icp_kernel over this rank's share.
    Returns the report (rank 0; None elsewhere).  The reference has no such mode: its `-batch` strip-mines ONE registration
    16 source rows at a time (src/libgpuicp/batchcovs.cu:28-75)."""
    import torch
    import icp_amd
    from icp_amd import synth
    per = (args.batch + world - 1) // world
    lo, hi = min(args.batch, rank * per), min(args.batch, (rank + 1) * per)
    n = args.batch_n
    P, Q = synth.make_batch(min(hi - lo, 64), n, seed=synth.SEED_Q + 1000 * rank)   # 64 distinct pairs, tiled
    reps = (hi - lo + len(P) - 1) // len(P)
    P, Q = np.tile(P, (reps, 1, 1))[:hi - lo], np.tile(Q, (reps, 1, 1))[:hi - lo]
    b = icp_amd.Batch(hi - lo, n, n, device=local_rank)
    b.upload(P, Q)
    have_pg = False
    if world > 1:
        import torch.distributed as td
        have_pg = td.is_initialized()

    def barrier():
        torch.cuda.synchronize()
        if have_pg:
            td.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        b.run(args.batch_iters)
    barrier()
    t0 = time.perf_counter()
    kms = 0.0
    for _ in range(steps):
        kms += b.run(args.batch_iters, timed=True)
    barrier()
    dt = time.perf_counter() - t0
    if have_pg:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda" if td.get_backend() == "nccl" else "cpu")
        td.all_reduce(tmax, op=td.ReduceOp.MAX)
        dt = float(tmax.item())
    b.close()
    if rank != 0:
        return None
    pairs = float(args.batch) * n * n * args.batch_iters
    ach = FLOPS_PER_PAIR * float(hi - lo) * n * n * args.batch_iters / (kms / steps * 1e-3)
    issue_frac = None
    if os.environ.get("ICP_AMD_SEARCH", "") not in ("exact", "fma"):   # the default forms; the difference forms have other counts
        seeded = os.environ.get("ICP_AMD_BATCH_SEEDED", "1" if n >= 384 else "0") != "0"   # the library's own default
        instr = float(hi - lo) * n * n * (7.0 + (3.5 if seeded else 7.0) * (args.batch_iters - 1))
        issue_frac = instr / (kms / steps * 1e-3) / (CUS * 4 * 16 * CLOCK_HZ)
    cpu = None
    if with_cpu:
        # the oracle port (the reference's loop, oracle/icp_oracle.c) on whole registrations of the same batch: each
        # registration is independent, so the host cores take one each (OpenMP inside one 1k x 1k search does not pay)
        from concurrent.futures import ThreadPoolExecutor
        from oracle import pyoracle as orc
        cores = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "0") or 0) or (os.cpu_count() or 1))
        regs = max(64, 4 * cores)
        t1 = time.perf_counter()
        with ThreadPoolExecutor(cores) as ex:     # ctypes releases the GIL inside the C call
            list(ex.map(lambda k: orc.icp(P[k % len(P)], Q[k % len(Q)], args.batch_iters, nthreads=1), range(regs)))
        cdt = time.perf_counter() - t1
        cpu = {"value": regs * float(n) * n * args.batch_iters / cdt, "unit": "distance evals/s", "cores": cores, "kind": "port",
               "registrations_per_sec": regs / cdt,
               "sample": "%d whole registrations (%d x %d, %d iterations) of the same batch, one per host thread, %.1f s"
                         % (regs, n, n, args.batch_iters, cdt)}
    batch_traffic = measured_traffic("batch_icp_kernel", n, n, world)
    try:   # the PMC pass must also be of this batch size and iteration count
        rec = json.load(open(os.path.join(ROOT, "icp_amd", "measured_traffic.json"))).get("batch_icp_kernel") or {}
        if batch_traffic[0] and (rec.get("batch"), rec.get("iters")) != (args.batch, args.batch_iters):
            batch_traffic = (None, "no PMC pass on this batch size / iteration count", None)
    except (OSError, ValueError):
        batch_traffic = (None, "no measured_traffic.json", None)
    return {
        "metric": "distance_evals_per_sec", "value": pairs * steps / dt, "unit": "distance evals/s",
        "registrations_per_sec": args.batch * steps / dt, "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": dt / steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": "BASELINE configs[4]: %d registrations of %d x %d points, %d iterations each, %d per GPU"
                               % (args.batch, n, n, args.batch_iters, hi - lo), "parallelism": "batch-split%d" % world},
        "roofline": {"bound": "valu_fp64", "kernel": "batch_icp_kernel", "achieved": ach / 1e12, "peak": PEAK_FP64_VALU_NOFMA / 1e12,
                     "unit": "TFLOP/s", "frac": ach / PEAK_FP64_VALU_NOFMA, "avg_launch_ms": kms / steps,
                     # one launch reads both clouds once and writes the registered cloud, the solve record and the
                     # indices; everything else stays in LDS / registers.  Algorithmic bytes, not a counter reading.
                     "algorithmic_bytes": float(hi - lo) * (n * 24 * 2 + n * 24 + n * 4 + 28 * 8),
                     "traffic": batch_traffic[0], "traffic_source_commit": batch_traffic[2], "traffic_note": batch_traffic[1],
                     "issue_frac": issue_frac,
                     "instruction_mix": "first iteration: 7 VALU instructions per pair (3 fp64 FMA, index embedded in the mantissa, "
                                        "best and runner-up); from the second on, seeded by the previous match: 3.5 (3 fp64 FMA + half "
                                        "a v_min3_u32 on the high words), candidate blocks re-scanned in the oracle's arithmetic",
                     "note": "8 un-fused fp64 flops per pair against the un-fused fp64 VALU peak; the default form issues 3.5 "
                             "instructions per pair from the second iteration on, so frac passes 1.  `issue_frac` = the fast "
                             "path's instruction count of this launch / its duration / the VALU issue rate at the nominal 2.4 GHz"},
        "cpu_baseline": cpu,
        "gpu_over_cpu": (pairs * steps / dt) / cpu["value"] if cpu else None,
        "reference_semantics": "no counterpart: the reference's -batch flag strip-mines ONE registration 16 source rows at a "
                               "time (src/libgpuicp/batchcovs.cu:28-75); here a batch is thousands of independent registrations, "
                               "one workgroup each, every one bit-identical to icp_amd_run on the same pair"}


def bench_batch(args, rank, world, local_rank):
    """`--workload batch`: BASELINE configs[4] as the whole JSON line (the default line carries it as `batch_mode`)."""
    out = measure_batch(args, rank, world, local_rank, args.steps, args.warmup, world == 1 and not args.no_cpu_baseline)
    if rank == 0:
        print(json.dumps(out), flush=True)
    import torch.distributed as td
    if td.is_available() and td.is_initialized():
        td.barrier()
        td.destroy_process_group()


def launch_ranks(n):
    """Runs this same command line under torch.distributed.run with `n` ranks on 127.0.0.1 and relays rank 0's JSON line.
    The child has a time limit of its own (ICP_AMD_BENCH_LAUNCH_TIMEOUT seconds, default 900): a rendezvous or an RCCL
    initialisation that never completes ends with the child's process group killed, what it printed so far on stderr and a
    non-zero exit, instead of running silently into whatever limit the caller has."""
    import signal
    import socket
    import subprocess
    import threading
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    limit = float(os.environ.get("ICP_AMD_BENCH_LAUNCH_TIMEOUT", "900"))
    # a fresh child in a session of its own (never a re-exec of this process), so that the whole tree can be signalled
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, start_new_session=True)
    lines, seen = [], []

    def relay():
        for ln in proc.stdout:
            if ln.startswith("{"):
                lines.append(ln)
            else:
                seen.append(ln)
                sys.stderr.write(ln)

    reader = threading.Thread(target=relay, daemon=True)
    reader.start()
    try:
        rc = proc.wait(timeout=limit)
    except subprocess.TimeoutExpired:
        for sig in (signal.SIGTERM, signal.SIGKILL):
            try:
                os.killpg(proc.pid, sig)
            except ProcessLookupError:
                break
            try:
                proc.wait(timeout=10)
                break
            except subprocess.TimeoutExpired:
                continue
        reader.join(timeout=5)
        sys.stderr.write("bench.py: the %d-rank child did not finish within %.0f s and was killed; it had printed %d line(s)%s\n"
                         % (n, limit, len(seen) + len(lines), (", the last: " + (lines + seen)[-1].strip()[:300]) if (lines or seen) else ""))
        raise SystemExit(124)
    reader.join(timeout=5)
    if lines:
        sys.stdout.write(lines[-1])
        sys.stdout.flush()
    if rc != 0:
        raise SystemExit(rc)
    if not lines:
        raise SystemExit("bench.py: the %d-rank child printed no JSON line" % n)
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--src", dest="n", type=int, default=1000000, help="source points (global)")
    ap.add_argument("--tgt", dest="m", type=int, default=1000000, help="target points")
    ap.add_argument("--search", default=os.environ.get("ICP_AMD_SEARCH", "f32"),
                    help="search variant of the headline measurement: f32 (default), fma, exact = brute force over all pairs")
    ap.add_argument("--no-grid", action="store_true", help="skip the extra grid-search measurement")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-batch", action="store_true", help="skip the configs[4] measurement (`batch_mode`) of the default line")
    ap.add_argument("--no-fp64-variants", action="store_true", help="skip the extra fp64-only search variants at N=1")
    ap.add_argument("--workload", default="sharded", choices=["sharded", "batch"],
                    help="sharded: BASELINE configs[3] (default, the headline metric); batch: configs[4], 4096 x (1k x 1k)")
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--batch-n", type=int, default=1000)
    ap.add_argument("--batch-iters", type=int, default=20)
    ap.add_argument("--n1-value", type=float, default=None,
                    help="`value` of the N=1 run of the same workload: adds scaling_vs_n1 = value / (n_gpus x that) to the line")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` (the shape of the driver's N=1 command): start one rank per GPU as a CHILD process,
        # before this process has imported torch or touched the GPU (a process that has initialised HIP is never re-exec'd),
        # relay its output and exit with its code.
        return launch_ranks(args.gpus)

    import torch
    import icp_amd
    from icp_amd import dist as icpdist
    from icp_amd import synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # rehearsal knobs (tests only): several ranks on one GPU over gloo, e.g. ICP_AMD_BENCH_DEVICE=0 ICP_AMD_BENCH_BACKEND=gloo
    if os.environ.get("ICP_AMD_BENCH_DEVICE"):
        local_rank = int(os.environ["ICP_AMD_BENCH_DEVICE"])
    backend = os.environ.get("ICP_AMD_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    # ICP_AMD_BENCH_FORCE_PG=1 (one rank, no launcher): a one-rank RCCL process group is created anyway and the iteration's
    # two all-gathers go through it, on the engine's stream, exactly as with eight ranks -- the one-GPU box's way of running
    # the real exchange (tests/test_gpu_sharded.py)
    force_pg = world == 1 and os.environ.get("ICP_AMD_BENCH_FORCE_PG", "0") == "1"
    pg = world > 1 or force_pg          # a process group exists: the exchanges and the bench's own reductions use it
    if pg:
        import torch.distributed as td
        if force_pg:
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                port = sk.getsockname()[1]
            td.init_process_group(backend, init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1,
                                  **({"device_id": torch.device("cuda", local_rank)} if backend == "nccl" else {}))
        elif backend == "nccl":
            td.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            td.init_process_group(backend)

    if args.workload == "batch":
        return bench_batch(args, rank, world, local_rank)

    N, M = args.n, args.m
    P, Q = synth.make_pair(N, M)          # identical on every rank (seeded)

    def barrier():
        torch.cuda.synchronize()
        if pg:
            td.barrier()
        torch.cuda.synchronize()

    def all_ranks(x):
        """[x of rank 0, x of rank 1, ...] as floats (the value itself without a process group)."""
        if not pg:
            return [float(x)]
        buf = torch.zeros(world, dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        buf[rank] = float(x)
        td.all_reduce(buf, op=td.ReduceOp.SUM)
        return [float(v) for v in buf.cpu()]

    def run(search, steps, warmup, repeats=0, extras=False, block_times=False, kernel_timing=True, block=10, sort_source=False):
        """`steps` timed iterations after `warmup` untimed ones.  Returns a dict: seconds (max over ranks), the search
        kernel's average launch ms (this rank's and every rank's), geometry, the last iteration's outputs, ...
        block_times (the grid legs, whose iteration is tens of microseconds): the iterations run ten at a time through
        ShardedIcp.steps -- on one rank the engine's own loop, one call across the C ABI per ten iterations."""
        runner = icpdist.ShardedIcp(P, Q, rank=rank, world=world, device=local_rank, search=search, kernel_timing=kernel_timing,
                                    force_collective=force_pg, sort_source=sort_source)
        runner.begin()
        first_ms = None
        for w in range(warmup):
            if w == 0:
                # the first iteration of a run has no previous correspondences to seed the search with: timed on its own,
                # outside the timed region, and reported as `first_step_ms`
                barrier()
                t1 = time.perf_counter()
                runner.step()
                barrier()
                first_ms = max(all_ranks((time.perf_counter() - t1) * 1e3))
            elif not block_times:
                runner.step()
        if block_times and warmup > 1:
            runner.steps(warmup - 1)
        barrier()
        runner.engine.kernel_ms()             # drop the warm-up launches from the kernel timer
        t0 = time.perf_counter()
        blocks = []
        if block_times:   # the grid legs: how the step time develops while the clouds converge
            for k in range(0, steps, block):
                runner.steps(min(block, steps - k))
                barrier()
                blocks.append(time.perf_counter())
        else:
            for k in range(steps):
                runner.step()
        barrier()
        dt = time.perf_counter() - t0
        dt = max(all_ranks(dt))
        avg_ms, _, launches = runner.engine.kernel_ms()
        res = {"dt": dt, "avg_ms": avg_ms, "launches": launches, "geo": runner.engine.search_geometry(),
               "last": runner.engine.get_last(), "n_local": runner.n_local, "grid_stats": runner.engine.grid_stats(),
               "first_ms": first_ms, "rank_ms": all_ranks(avg_ms),
               "ms_per_step_by_10": [round((b - a) * 1e3 / block, 4) for a, b in zip([t0] + blocks[:-1], blocks)] if block == 10 else None}
        # outside the timed region: a few more steps, each bracketed on its own, for the spread (median + min)
        for _ in range(repeats):
            barrier()
            t1 = time.perf_counter()
            runner.step()
            barrier()
            step_samples.append((time.perf_counter() - t1) * 1e3)
        if extras and pg:
            # the two exchanges of an iteration on their own (what the ranks wait for besides their kernels): `reps`
            # all-gathers of each buffer back to back on the runner's stream, bracketed like a step
            reps = 50
            res["collective_ms"] = {}
            for name, buf, width in (("cov_partials", runner.cov, icpdist.COV_W), ("centroid_partials", runner.cen, icpdist.CEN_W)):
                keep = buf.clone()
                barrier()
                t1 = time.perf_counter()
                with runner._on_stream():
                    for _ in range(reps):
                        icpdist.exchange_partials(buf, width, rank, world, force=True)
                barrier()
                res["collective_ms"][name] = {"ms": max(all_ranks((time.perf_counter() - t1) * 1e3 / reps)), "bytes": buf.numel() * 8}
                buf.copy_(keep)
            # SURVEY.md 8e's debug cross-check: every rank solved R, t redundantly from the same gathered partials, so rank 0's
            # R || t, broadcast, must equal every rank's own bit for bit (always on for N > 1; ICP_AMD_BENCH_CROSSCHECK=0 skips)
            if os.environ.get("ICP_AMD_BENCH_CROSSCHECK", "1") != "0":
                mine = torch.from_numpy(np.concatenate([res["last"]["R"].ravel(), res["last"]["t"].ravel()]).copy())
                root = mine.clone().to("cuda" if backend == "nccl" else "cpu")
                td.broadcast(root, src=0)
                same = float(torch.equal(root.cpu(), mine))
                res["rt_crosscheck"] = {"ranks_identical_to_rank0": int(sum(all_ranks(same))), "of": world,
                                        "what": "R || t of the last iteration, broadcast from rank 0 and compared bit for bit"}
        runner.close()
        return res

    step_samples = []

    # Headline: the brute-force search (every one of the N x M distances is evaluated each iteration) -- the path the
    # metric "distance evals/s" and the VALU roofline are defined on.
    head = run(args.search, args.steps, args.warmup, repeats=5, extras=True)
    dt, search_avg_ms, launches, geo, last, n_local = (head[k] for k in ("dt", "avg_ms", "launches", "geo", "last", "n_local"))
    grid = None
    if not args.no_grid and args.search != "grid":
        # The same iterations with the grid search (identical results, a fraction of the distance evaluations):
        # reported beside the headline as iterations/s, never as evals/s.
        g_steps = max(args.steps * 10, 50)
        # two measurements: the first g_steps iterations of a registration (the clouds still move, search boxes are wide,
        # few rows hold a motion certificate) and g_steps iterations after 3 x g_steps more (the registration has settled:
        # what a long run spends most of its iterations in)
        # (step times from engines without per-launch events -- two event records are 10 % of a 50 us iteration -- and the
        # search kernel's own time from a second pass with them)
        early = run("grid", g_steps, max(args.warmup, 3), block_times=True, kernel_timing=False)
        # (the settled leg as ONE call of g_steps iterations: a barrier + synchronize every ten iterations, which the early leg
        # pays to show how the step time develops, costs a 50 us iteration 3-6 % in refill gaps)
        gr = run("grid", g_steps, max(args.warmup, 3) + 3 * g_steps, block_times=True, kernel_timing=False, block=g_steps)
        early["avg_ms"] = run("grid", g_steps, max(args.warmup, 3), block_times=True)["avg_ms"]
        gr["avg_ms"] = run("grid", g_steps, max(args.warmup, 3) + 3 * g_steps, block_times=True)["avg_ms"]
        g_dt, g_ms, g_geo, g_stats = gr["dt"], gr["avg_ms"], gr["geo"], gr["grid_stats"]
        grid = {"ms_per_step": g_dt / g_steps * 1e3, "iters_per_sec": g_steps / g_dt, "steps": g_steps,
                "warmup_steps": max(args.warmup, 3) + 3 * g_steps,
                "speedup_vs_brute_force": (dt / args.steps) / (g_dt / g_steps), "search_kernel_avg_ms": g_ms,
                "search": g_geo, "stats": g_stats, "first_step_ms": gr["first_ms"],
                # HBM roofline of the settled grid search (hipEvent average over the timed launches).  Since round 3 that launch is
                # search_grid_settled_kernel, which searches AND forms the covariance sums: 100 algorithmic bytes per source row --
                # the row's two 32-byte record halves in, the match half out, 4 bytes of index out
                "roofline": grid_roofline(gr["n_local"], g_ms, g_dt / g_steps, world),
                "first_steps": {"ms_per_step": early["dt"] / g_steps * 1e3, "iters_per_sec": g_steps / early["dt"], "steps": g_steps,
                                "warmup_steps": max(args.warmup, 3), "ms_per_step_by_10": early["ms_per_step_by_10"],
                                "search_kernel_avg_ms": early["avg_ms"]},
                "note": "exact nearest neighbour over a uniform grid (icp_amd/csrc/grid.hip): same correspondences as the "
                        "brute-force search, only the targets near each source point are evaluated; not counted in `value`. "
                        "`ms_per_step` is taken once the registration has settled (after `warmup_steps` iterations); "
                        "`first_steps` is the same measurement at the start of the registration"}
        if world == 1 and not force_pg:
            # the same first g_steps iterations with the source STORED in spatial order (icp_amd_options.sort_source; off by
            # default because the reduction order, hence the last bits of Pc / cov, then follows the stored rows)
            se = run("grid", g_steps, max(args.warmup, 3), block_times=True, kernel_timing=False, sort_source=True)
            grid["sorted_source"] = {
                "first_steps": {"ms_per_step": se["dt"] / g_steps * 1e3, "steps": g_steps, "ms_per_step_by_10": se["ms_per_step_by_10"]},
                "rms_last_relative_difference": abs(se["last"]["rms"] - early["last"]["rms"]) / early["last"]["rms"],
                # (after ~100 iterations on clouds that differ in the last bits: a row between two targets an ulp apart may differ)
                "rows_whose_index_differs_from_unsorted": int((se["last"]["idx"] != early["last"]["idx"]).sum()),
                "note": "icp_amd_options.sort_source = 1: rows sorted by a coarse cell at set_source, clouds and indices in and out "
                        "in the caller's order; the settled iteration is unchanged (it streams either way)"}

    # BASELINE configs[4] beside the headline (one warm-up + three launches of 4096 x (1k x 1k) x 20 iterations, ~0.1 s of GPU
    # time; its CPU leg -- the oracle port on whole registrations, all cores -- a few seconds): `--no-batch` skips it
    batch_mode = None
    if not args.no_batch:
        batch_mode = measure_batch(args, rank, world, local_rank, 3, 1, world == 1 and not args.no_cpu_baseline)

    if rank == 0:
        pairs = float(N) * float(M)
        evals_per_s = pairs * args.steps / dt
        out = {
            "metric": "distance_evals_per_sec", "value": evals_per_s, "unit": "distance evals/s",
            "iters_per_sec": args.steps / dt,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s: synthetic %d x %d points, source rows sharded over %d GPU(s), target replicated"
                                   % ("BASELINE configs[3]" if N == M == 1000000 else "BASELINE configs[2]" if N == M == 100000
                                      else "configs[3]-shaped workload at another size", N, M, world),
                       "n": N, "m": M, "parallelism": "p-shard%d" % world, "search": geo,
                       "search_mode": "brute force: all N x M distances evaluated every iteration",
                       "scaling_note": "north_star's >= 6x from 1 to 8 GPUs applies to THIS path (brute force: 27 ms of search per "
                                       "iteration at N = 1 against two latency-bound all-gathers); the grid search reported under "
                                       "`grid_search` computes for ~40 us per iteration at 8 GPUs and is bound by those two "
                                       "collectives, so sharding it buys little and no 6x is claimed for it"},
            # dominant kernel (the correspondence search, > 99 % of the step), from its own hipEvent-timed launches
            "roofline": roofline(geo["variant"], n_local, M, search_avg_ms, launches),
            "rms_last": last["rms"],
            "first_step_ms": head["first_ms"],
            "search_ms_per_rank": {"max": max(head["rank_ms"]), "min": min(head["rank_ms"]),
                                   "values": [round(v, 4) for v in head["rank_ms"]]},
            "process_group": {"ranks": td.get_world_size() if pg else 1,
                              "backend": (("rccl (torch 'nccl')" if backend == "nccl" else backend) if pg else "none"),
                              "forced_single_rank": bool(force_pg)},
            # ranks of the RCCL communicator the iteration's all-gathers went through (0: no RCCL call was issued)
            "rccl_ranks": (td.get_world_size() if (pg and backend == "nccl") else 0),
            "step_ms_samples": {"values": [round(v, 4) for v in step_samples], "median": float(np.median(step_samples)),
                                "min": float(min(step_samples)),
                                "note": "five more iterations after the timed region, each bracketed by its own barrier + "
                                        "synchronize (rank 0's clock); not part of `value`"},
        }
        if geo["variant"] == "f32":
            tb, src, blob = measured_traffic(out["roofline"]["kernel"], N, M, world)
            out["roofline"]["traffic"] = tb
            out["roofline"]["traffic_source_commit"] = blob
            if not tb:
                out["roofline"]["traffic_note"] = src
            if tb:
                out["roofline"]["traffic_note"] = (
                    "bytes per launch at the L2's memory side (FETCH_SIZE x 2 + WRITE_SIZE, separate rocprofv3 --pmc passes: %s); "
                    "includes Infinity-Cache hits; the kernel is not memory bound: this is %.2f TB/s"
                    % (src, tb / (search_avg_ms * 1e-3) / 1e12))
        out["roofline"]["first_step_ms"] = head["first_ms"]
        if "collective_ms" in head:
            out["collective_ms"] = head["collective_ms"]
        if "rt_crosscheck" in head:
            out["rt_crosscheck"] = head["rt_crosscheck"]
        if args.n1_value:
            out["scaling_vs_n1"] = out["value"] / (world * args.n1_value)
            out["speedup_vs_n1"] = out["value"] / args.n1_value
        if grid:
            out["grid_search"] = grid
        if batch_mode:
            out["batch_mode"] = batch_mode
        if world == 1 and not args.no_fp64_variants:
            out["fp64_variants"] = fp64_variants(P, Q, local_rank)
            # the roofline SURVEY.md 8(d) defines: 8 un-fused fp64 flops per pair against the un-fused fp64 VALU peak, for the
            # two variants whose fast path IS fp64 VALU work (same indices as the headline kernel, same run)
            for var in ("exact", "fma"):
                out["roofline"]["frac_fp64_valu_" + var] = out["fp64_variants"][var]["roofline"]["frac"]
            out["parity_check"] = parity_check(P, Q, local_rank)
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(P, Q)
            out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
            ref = cpu_reference(P, Q)
            if ref:
                out["cpu_reference"] = ref
        print(json.dumps(out), flush=True)
    if pg:
        td.barrier()
        td.destroy_process_group()


if __name__ == "__main__":
    main()
