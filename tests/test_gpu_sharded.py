"""GPU: the P-sharded path on a one-GPU box.  W engines (rank r of world W) run in one process on cuda:0 and the
all-gathers of the per-chunk partials are emulated by device copies between the engines' exchange buffers.
Results must be bit-identical to the single-engine run (fixed-order reductions make them independent of W)."""
import numpy as np
import pytest

import icp_amd
from icp_amd import dist as icpdist
from icp_amd import synth

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def sticky_hip_error():
    """hipPeekAtLastError() of this host thread: the process-wide sticky error word the library must neither leave dirty
    (torch would report the leftover as its own failure) nor clear on behalf of others."""
    import ctypes
    return int(ctypes.CDLL("libamdhip64.so").hipPeekAtLastError())


def run_sharded(P, Q, world, iters):
    n = len(P)
    cpr = icp_amd.chunks_per_rank(n, world)
    dev = torch.device("cuda", 0)
    tstream = torch.cuda.Stream(dev)      # torch's default stream has handle 0 = "engine, make your own"
    stream = tstream.cuda_stream
    covs = [torch.zeros(world * cpr * 12, dtype=torch.float64, device=dev) for _ in range(world)]
    cens = [torch.zeros(world * cpr * 3, dtype=torch.float64, device=dev) for _ in range(world)]
    torch.cuda.synchronize()
    engines = []
    for r in range(world):
        lo, hi = icpdist.shard_bounds(n, r, world)
        e = icp_amd.Engine(stream=stream, rank=r, world=world, n_global=n, cov_gather=covs[r].data_ptr(),
                           cen_gather=cens[r].data_ptr())
        e.set_target(Q)
        e.set_source(np.ascontiguousarray(P[lo:hi]))
        engines.append(e)

    def allgather(bufs, width):
        per = cpr * width
        with torch.cuda.stream(tstream):
            for r in range(world):
                for o in range(world):
                    if o != r:
                        bufs[o][r * per:(r + 1) * per].copy_(bufs[r][r * per:(r + 1) * per])

    for e in engines:
        e.step_begin()
    allgather(cens, 3)
    lasts = None
    for _ in range(iters):
        for e in engines:
            e.step_search()
        allgather(covs, 12)
        for e in engines:
            e.step_update()
        allgather(cens, 3)
    torch.cuda.synchronize()
    lasts = [e.get_last() for e in engines]
    P_out = np.concatenate([e.get_source() for e in engines])
    for e in engines:
        e.close()
    return lasts, P_out


@pytest.mark.parametrize("n,m,world", [(20000, 20000, 2), (5000, 7000, 3), (100000, 100000, 8), (1500, 1500, 4)])
def test_sharded_equals_single(n, m, world):
    P, Q = synth.make_pair(n, m, seed=11 + world)
    iters = 3
    e = icp_amd.Engine()
    e.set_target(Q)
    e.set_source(P)
    e.iterate(iters)
    single = e.get_last()
    P_single = e.get_source()
    e.close()
    lasts, P_out = run_sharded(P, Q, world, iters)
    assert np.array_equal(np.concatenate([l["idx"] for l in lasts]), single["idx"])
    for l in lasts:                       # every rank solves redundantly: identical R, t, cov everywhere
        for k in ("cov", "R", "t", "Pc", "S"):
            assert np.array_equal(l[k], single[k]), k
    assert np.array_equal(P_out, P_single)


def test_torch_stream_and_single_rank_process_group():
    """world = 1 through the same code path bench.py uses under torchrun (caller-owned exchange buffers on
    torch's current stream); the all-gather of a one-rank group is the identity."""
    P, Q = synth.make_pair(3000, 3000, seed=5)
    dev = torch.device("cuda", 0)
    cpr = icp_amd.chunks_per_rank(len(P), 1)
    cov = torch.zeros(cpr * 12, dtype=torch.float64, device=dev)
    cen = torch.zeros(cpr * 3, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    tstream = torch.cuda.Stream(dev)
    e = icp_amd.Engine(stream=tstream.cuda_stream, cov_gather=cov.data_ptr(), cen_gather=cen.data_ptr())
    e.set_target(Q)
    e.set_source(P)
    e.step_begin()
    for _ in range(2):
        e.step_search()
        e.step_update()
    torch.cuda.synchronize()
    a = e.get_last()
    # the covariance the engine solved is the in-order sum of the chunk partials it left in the caller's buffer
    parts = cov.cpu().numpy().reshape(cpr, 12)
    acc = np.zeros(12)
    for r in range(cpr):
        acc = acc + parts[r]
    assert np.array_equal(acc[:9].reshape(3, 3), a["cov"])
    e.close()
    b_out, tr = icp_amd.icp_gpu_optimized(P, Q, 2, trace=True)
    assert np.array_equal(tr["idx"][1], a["idx"]) and np.array_equal(tr["cov"][1], a["cov"])


def test_bench_two_ranks_on_one_gpu_over_gloo():
    """bench.py's multi-rank path end to end, started the way the driver starts it -- `python bench.py --gpus 2` with no
    torchrun around it: bench.py launches its own torch.distributed.run child (2 processes, ShardedIcp with caller-owned
    exchange buffers on its own stream) before it touches the GPU, and relays rank 0's line.  Rehearsed on the one GPU of
    this box with the gloo backend standing in for RCCL."""
    import json
    import os
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, ICP_AMD_BENCH_DEVICE="0", ICP_AMD_BENCH_BACKEND="gloo", OMP_NUM_THREADS="4")
    env.pop("WORLD_SIZE", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--src", "60000", "--tgt", "50000", "--n1-value", "1e12"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["config"]["parallelism"] == "p-shard2"
    assert d["process_group"] == {"ranks": 2, "backend": "gloo", "forced_single_rank": False}
    assert d["rccl_ranks"] == 0                                       # gloo rehearsal: no RCCL call was issued
    assert len(d["search_ms_per_rank"]["values"]) == 2 and d["search_ms_per_rank"]["min"] > 0
    assert set(d["collective_ms"]) == {"cov_partials", "centroid_partials"}
    assert d["rt_crosscheck"]["ranks_identical_to_rank0"] == 2      # every rank solved the same R, t, bit for bit
    assert d["first_step_ms"] > 0 and abs(d["scaling_vs_n1"] - d["value"] / 2e12) < 1e-12
    assert d["config"]["search"]["variant"] == "f32" and d["grid_search"]["search"]["variant"] == "grid"
    assert d["grid_search"]["iters_per_sec"] > 0
    # same workload on one rank: the registration must reach the same rms (bit-identical results whatever the sharding)
    r1 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--src", "60000", "--tgt", "50000",
                         "--no-cpu-baseline", "--no-fp64-variants"], env=env, capture_output=True, text=True, timeout=280)
    assert r1.returncode == 0, r1.stderr[-3000:]
    d1 = json.loads([ln for ln in r1.stdout.splitlines() if ln.startswith("{")][-1])
    assert d1["rms_last"] == d["rms_last"]


def test_bench_batch_workload_split_over_two_ranks_over_gloo():
    """BASELINE configs[4]'s multi-GPU leg in rehearsal: `bench.py --gpus 2 --workload batch` splits the registrations evenly
    over the ranks (no data-path collective: a barrier and a MAX over the ranks' times are all that is exchanged), started
    the way the driver starts it; two ranks on the one GPU of this box, gloo standing in for RCCL."""
    import json
    import os
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, ICP_AMD_BENCH_DEVICE="0", ICP_AMD_BENCH_BACKEND="gloo", OMP_NUM_THREADS="4")
    env.pop("WORLD_SIZE", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "batch", "--batch", "96", "--batch-n", "500",
           "--batch-iters", "5", "--steps", "2", "--warmup", "1"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["config"]["parallelism"] == "batch-split2"
    assert "48 per GPU" in d["config"]["workload"] and d["scaling"] == "strong"
    # whole-job figures: all 96 registrations over the slowest rank's time
    assert abs(d["registrations_per_sec"] - 96 * 2 / (d["ms_per_step"] * 2e-3)) < 1e-6 * d["registrations_per_sec"]
    assert abs(d["value"] - d["registrations_per_sec"] * 500 * 500 * 5) < 1e-6 * d["value"]
    assert d["roofline"]["kernel"] == "batch_icp_kernel" and 0 < d["roofline"]["frac"] < 1.2 and d["cpu_baseline"] is None


def test_bench_real_rccl_exchange_with_a_one_rank_process_group():
    """The exchange the multi-GPU benchmark relies on, executed for real on the one GPU of this box: ICP_AMD_BENCH_FORCE_PG=1
    makes bench.py create an RCCL process group of ONE rank (torch backend "nccl", device_id bound) and send the two
    all-gathers of every iteration through it -- td.all_gather_into_tensor on the engine's stream, the CUDA-tensor
    all_reduce / broadcast of the bench's own bookkeeping -- exactly the calls eight ranks issue.  Same registration as the
    plain run (no process group, no collective): identical rms, bit for bit."""
    import json
    import os
    import subprocess
    import sys
    from conftest import ROOT
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--src", "60000", "--tgt", "50000",
            "--no-cpu-baseline", "--no-fp64-variants", "--no-batch"]
    env = dict(os.environ, OMP_NUM_THREADS="4")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "ICP_AMD_BENCH_BACKEND", "ICP_AMD_BENCH_DEVICE"):
        env.pop(k, None)
    out = {}
    for name, extra in (("forced", {"ICP_AMD_BENCH_FORCE_PG": "1"}), ("plain", {})):
        r = subprocess.run(base, env=dict(env, **extra), capture_output=True, text=True, timeout=280)
        assert r.returncode == 0, (name, r.stdout[-2000:] + r.stderr[-4000:])
        out[name] = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    d, d1 = out["forced"], out["plain"]
    assert d["process_group"] == {"ranks": 1, "backend": "rccl (torch 'nccl')", "forced_single_rank": True}
    assert d["rccl_ranks"] == 1 and d["n_gpus"] == 1
    assert set(d["collective_ms"]) == {"cov_partials", "centroid_partials"}
    assert all(v["ms"] > 0 and v["bytes"] > 0 for v in d["collective_ms"].values())
    assert d["rt_crosscheck"] == dict(d["rt_crosscheck"], ranks_identical_to_rank0=1, of=1)
    assert d1["process_group"]["backend"] == "none" and d1["rccl_ranks"] == 0 and "collective_ms" not in d1
    assert d["rms_last"] == d1["rms_last"] and d["rms_last"] > 0
    assert d["grid_search"]["iters_per_sec"] > 0 and d["config"]["search"]["variant"] == "f32"


def test_bench_child_launch_has_a_time_limit_of_its_own():
    """`python bench.py --gpus 2` starts its ranks as a child process; a child that does not finish within
    ICP_AMD_BENCH_LAUNCH_TIMEOUT seconds is killed (its whole process group), what it had printed is reported and the exit
    status is non-zero -- a hung RCCL initialisation no longer runs silently into the caller's own limit."""
    import os
    import subprocess
    import sys
    import time
    from conftest import ROOT
    env = dict(os.environ, ICP_AMD_BENCH_DEVICE="0", ICP_AMD_BENCH_BACKEND="gloo", ICP_AMD_BENCH_LAUNCH_TIMEOUT="1", OMP_NUM_THREADS="4")
    env.pop("WORLD_SIZE", None)
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 124, (r.returncode, r.stderr[-2000:])
    assert "did not finish within 1 s and was killed" in r.stderr
    assert time.time() - t0 < 60


def test_native_multi_cross_check_of_the_redundant_solves(monkeypatch):
    """ICP_AMD_MULTI_CROSSCHECK=1 (SURVEY 8e's debug aid for the first real multi-GPU run): shard 0's R || t is broadcast
    (ncclBroadcast on a one-rank communicator here; peer copies on a device list that repeats device 0) and compared bit for
    bit with every shard's own redundant solve.  Results are unchanged and the check passes."""
    monkeypatch.setenv("ICP_AMD_MULTI_CROSSCHECK", "1")
    P, Q = synth.make_pair(9000, 8000, seed=47)
    want = icp_amd.icp_gpu_optimized(P, Q, 4)
    assert sticky_hip_error() == 0
    assert np.array_equal(icp_amd.icp_gpu_multi(P, Q, 4, 1, devices=[0]), want)            # RCCL, one rank
    assert sticky_hip_error() == 0            # RCCL's device probing leaves no stale error behind our entry point
    assert np.array_equal(icp_amd.icp_gpu_multi(P, Q, 4, 3, devices=[0, 0, 0]), want)      # peer-copy exchange, three shards
    assert sticky_hip_error() == 0
    torch.zeros(4, device="cuda").sum().item()   # ... so the next HIP user of the process (torch) is not handed one


def test_native_rccl_multi_entry_point():
    """icp_amd_run_multi (single process, RCCL loaded at run time).  On this one-GPU box: a one-rank RCCL communicator
    on an explicit device list exercises the load / init / in-place all-gather / teardown path, and must equal icp_amd_run;
    asking for more GPUs than are visible is a clean error."""
    P, Q = synth.make_pair(6000, 5000, seed=41)
    want, trw = icp_amd.icp_gpu_optimized(P, Q, 3, trace=True)
    got, trg = icp_amd.icp_gpu_multi(P, Q, 3, 1, devices=[0], trace=True)
    assert np.array_equal(got, want)
    for k in ("idx", "cov", "R", "t", "Pc"):
        assert np.array_equal(trg[k], trw[k]), k
    assert np.array_equal(icp_amd.icp_gpu_multi(P, Q, 3, 1), want)
    with pytest.raises(icp_amd.IcpAmdError):
        icp_amd.icp_gpu_multi(P, Q, 1, icp_amd.device_count() + 1)


def test_native_multi_entry_point_several_shards_per_gpu(monkeypatch):
    """icp_amd_run_multi with a device list that repeats device 0: the whole sharded control flow of the native entry point
    on this one-GPU box -- shard arithmetic, one set-up thread per shard, engines sharing a device, the exchange (peer
    copies ordered by events, because RCCL refuses a device listed twice), per-shard trace indices, empty trailing
    shards.  Bit-identical to the single-engine run for every shard count."""
    for n, m, iters in ((6000, 5000, 4), (700, 900, 3), (20000, 20000, 3)):
        P, Q = synth.make_pair(n, m, seed=43)
        want, trw = icp_amd.icp_gpu_optimized(P, Q, iters, trace=True)
        for shards in (2, 3, 5):
            got, trg = icp_amd.icp_gpu_multi(P, Q, iters, shards, devices=[0] * shards, trace=True)
            assert np.array_equal(got, want), (n, shards)
            for k in ("idx", "cov", "R", "t", "Pc"):
                assert np.array_equal(trg[k], trw[k]), (n, shards, k)
    # the same exchange on a list RCCL could serve
    monkeypatch.setenv("ICP_AMD_MULTI_EXCHANGE", "copy")
    P, Q = synth.make_pair(6000, 5000, seed=44)
    assert np.array_equal(icp_amd.icp_gpu_multi(P, Q, 3, 1, devices=[0]), icp_amd.icp_gpu_optimized(P, Q, 3))
    monkeypatch.delenv("ICP_AMD_MULTI_EXCHANGE")
    with pytest.raises(icp_amd.IcpAmdError):
        icp_amd.icp_gpu_multi(P, Q, 1, 2, devices=[0, icp_amd.device_count()])
