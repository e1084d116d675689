"""P-sharded ICP across the GPUs of one node: one process per GPU (torch.distributed, backend "nccl" = RCCL
over xGMI).  The reference has no multi-GPU path; this is new.

Partitioning: rank r owns the source rows [r*rows_per_rank, (r+1)*rows_per_rank) (rows_per_rank a multiple of
the 1024-row reduction chunk); the target cloud is replicated.  Per iteration there are two tiny exchanges, both
all-gathers of per-chunk partial sums (12 doubles per chunk for the covariance, 3 for the centroid); every
rank then sums ALL chunks in the same fixed order and solves R, t redundantly, so results are bit-identical on every
rank and independent of the number of GPUs, and no broadcast of R, t is needed.
"""
import numpy as np

from . import Engine, chunks_per_rank, rows_per_rank

COV_W, CEN_W = 12, 3


def shard_bounds(n_global, rank, world):
    """[lo, hi) source rows owned by `rank`."""
    rpr = rows_per_rank(n_global, world)
    lo = min(n_global, rank * rpr)
    return lo, min(n_global, lo + rpr)


def exchange_partials(buf, width, rank, world, group=None, force=False):
    """All-gather of the rank-major partial array `buf` (world * chunks_per_rank * width doubles, torch tensor on
    any device): every rank contributes its own slot and receives all others.  Works on "nccl" and "gloo".
    world == 1 needs no exchange and issues none, unless `force` asks for the collective anyway (a one-rank process
    group: the same RCCL call, enqueued on the same stream, as with eight ranks -- how the one-GPU box exercises it)."""
    if world == 1 and not force:
        return buf
    import torch.distributed as td
    per = buf.numel() // world
    mine = buf[rank * per:(rank + 1) * per].clone()
    if td.get_backend(group) == "gloo":
        # gloo (tests, CPU rehearsals) has no device all-gather: stage through the host
        host = [mine.new_empty(per, device="cpu") for _ in range(world)]
        td.all_gather(host, mine.cpu(), group=group)
        for r in range(world):
            buf[r * per:(r + 1) * per].copy_(host[r])
    else:
        td.all_gather_into_tensor(buf, mine, group=group)
    return buf


class ShardedIcp:
    """Drives one engine per process.  world == 1 needs no torch at all."""

    def __init__(self, P, Q, rank=0, world=1, device=0, search="auto", kernel_timing=False, group=None, force_collective=False,
                 sort_source=False):
        self.rank, self.world, self.group = rank, world, group
        self.collective = world > 1 or force_collective   # exchange buffers + the two all-gathers per iteration
        n_global = len(P)
        lo, hi = shard_bounds(n_global, rank, world)
        self.lo, self.hi, self.n_local, self.n_global = lo, hi, hi - lo, n_global
        self.cov = self.cen = self.stream = None
        kw = dict(device=device, search=search, kernel_timing=kernel_timing, sort_source=sort_source)   # (sort_source: one rank only)
        if self.collective:
            import torch
            cpr = chunks_per_rank(n_global, world)
            dev = torch.device("cuda", device)
            self.cov = torch.zeros(world * cpr * COV_W, dtype=torch.float64, device=dev)
            self.cen = torch.zeros(world * cpr * CEN_W, dtype=torch.float64, device=dev)
            torch.cuda.synchronize(dev)
            # a stream of our own: torch's default stream has handle 0, which the C ABI reads as "create one", and
            # the collectives must be ordered with the engine's kernels (torch syncs its NCCL stream with the
            # stream that is current when the collective is issued)
            self.stream = torch.cuda.Stream(dev)
            kw.update(stream=self.stream.cuda_stream, rank=rank, world=world, n_global=n_global,
                      cov_gather=self.cov.data_ptr(), cen_gather=self.cen.data_ptr())
        self.engine = Engine(**kw)
        self.engine.set_target(Q)
        self.engine.set_source(np.ascontiguousarray(P[lo:hi]))

    def _on_stream(self):
        if self.stream is None:
            import contextlib
            return contextlib.nullcontext()
        import torch
        return torch.cuda.stream(self.stream)

    def begin(self):
        with self._on_stream():
            self.engine.step_begin()
            if self.collective:
                exchange_partials(self.cen, CEN_W, self.rank, self.world, self.group, force=True)

    def step(self):
        with self._on_stream():
            self.engine.step_search()
            if self.collective:
                exchange_partials(self.cov, COV_W, self.rank, self.world, self.group, force=True)
            self.engine.step_update()
            if self.collective:
                exchange_partials(self.cen, CEN_W, self.rank, self.world, self.group, force=True)

    def steps(self, k):
        """k iterations.  Without an exchange (one rank) that is the engine's own loop -- one call across the C ABI, as a host
        program would make it -- instead of three calls per iteration from Python."""
        if not self.collective:
            self.engine.iterate(k)
        else:
            for _ in range(k):
                self.step()

    def gather_source(self):
        """The registered cloud, assembled on every rank (host arrays)."""
        local = self.engine.get_source()      # synchronises the engine's stream
        if not self.collective:
            return local
        import torch
        import torch.distributed as td
        rpr = rows_per_rank(self.n_global, self.world)
        pad = torch.zeros((rpr, 3), dtype=torch.float64, device=self.cov.device)
        pad[:self.n_local] = torch.from_numpy(local).to(self.cov.device)
        out = torch.empty((self.world * rpr, 3), dtype=torch.float64, device=self.cov.device)
        td.all_gather_into_tensor(out, pad, group=self.group)
        return out[:self.n_global].cpu().numpy()

    def close(self):
        self.engine.close()
