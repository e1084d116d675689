"""Worker of tests/test_dist_cpu.py: one rank of a world_size-2 gloo job on CPU.  The per-shard arithmetic is done
by the CPU oracle (test infrastructure) so that the sharding protocol of icp_amd.dist -- row partition, all-gather of
rank-major chunk partials, in-order global sum, redundant solve -- is exercised end to end without a GPU."""
import os
import sys

import numpy as np
import torch
import torch.distributed as td

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import icp_amd  # noqa: E402
from icp_amd import dist as icpdist  # noqa: E402
from icp_amd import synth  # noqa: E402
from oracle import pyoracle as orc  # noqa: E402

CH = 1024


def chunk_sums(A, cpr, width):
    out = np.zeros((cpr, width))
    for c in range((len(A) + CH - 1) // CH):
        out[c, :A.shape[1]] = A[c * CH:(c + 1) * CH].sum(axis=0)
    return out


def ordered_total(buf, width):
    acc = np.zeros(width)
    for row in buf.reshape(-1, width):
        acc = acc + row
    return acc


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    td.init_process_group("gloo", rank=rank, world_size=world)
    n, m, iters = 5000, 3000, 3
    P, Q = synth.make_pair(n, m if m >= n else n, seed=21)
    Q = Q[:m] if m < n else Q
    lo, hi = icpdist.shard_bounds(n, rank, world)
    cpr = icp_amd.chunks_per_rank(n, world)
    Qc = orc.mean0(Q)
    Qz = orc.sub_row(Q, Qc)
    Pl = P[lo:hi].copy()
    cen = torch.zeros(world * cpr * 3, dtype=torch.float64)
    cov = torch.zeros(world * cpr * 12, dtype=torch.float64)

    def put(buf, width, arr):
        buf[rank * cpr * width:(rank + 1) * cpr * width] = torch.from_numpy(arr.reshape(-1))

    put(cen, 3, chunk_sums(Pl, cpr, 3))
    icpdist.exchange_partials(cen, 3, rank, world)
    idx_all = []
    for _ in range(iters):
        Pc = ordered_total(cen.numpy(), 3) / n
        Pz = Pl - Pc
        idx = orc.correspondences(Pz, Qz)
        idx_all.append(idx)
        prod = np.einsum("ia,ib->iab", Qz[idx], Pz).reshape(-1, 9)
        put(cov, 12, chunk_sums(prod, cpr, 12))
        icpdist.exchange_partials(cov, 12, rank, world)
        C = ordered_total(cov.numpy(), 12)[:9].reshape(3, 3)
        U, S, VT = orc.svd3(C)
        R, t = orc.rt(U, VT, Pc, Qc)
        Pl = orc.apply(Pl, R, t)
        put(cen, 3, chunk_sums(Pl, cpr, 3))
        icpdist.exchange_partials(cen, 3, rank, world)
    # every rank holds identical R, t: compare through a broadcast of rank 0's
    Rt = torch.from_numpy(np.concatenate([R.reshape(-1), t]))
    Rt0 = Rt.clone()
    td.broadcast(Rt0, 0)
    assert torch.equal(Rt, Rt0), "ranks disagree on R, t"
    # assemble the registered cloud on every rank and compare with the single-process oracle
    rpr = icp_amd.rows_per_rank(n, world)
    pad = torch.zeros((rpr, 3), dtype=torch.float64)
    pad[:hi - lo] = torch.from_numpy(Pl)
    parts = [torch.zeros((rpr, 3), dtype=torch.float64) for _ in range(world)]
    td.all_gather(parts, pad)
    P_out = torch.cat(parts)[:n].numpy()
    ref = orc.icp(P, Q, iters, trace=True)
    assert np.array_equal(np.stack(idx_all), ref["idx"][:, lo:hi]), "sharded correspondences differ"
    err = np.abs(P_out - ref["P_out"]).max() / np.abs(ref["P_out"]).max()
    assert err < 1e-9, err
    td.barrier()
    td.destroy_process_group()
    print("rank %d ok err %.2e" % (rank, err))


if __name__ == "__main__":
    main()
