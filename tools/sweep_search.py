#!/usr/bin/env python3
"""Sweep the search-kernel launch parameters on one GPU (one process, interleaved configs).
usage: python tools/sweep_search.py [n] [m] [iters]"""
import itertools
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import icp_amd  # noqa: E402
from icp_amd import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
m = int(sys.argv[2]) if len(sys.argv) > 2 else n
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 4
P, Q = synth.make_pair(n, m)
variants = os.environ.get("SWEEP_VARIANTS", "exact,fma").split(",")
cbs = [int(x) for x in os.environ.get("SWEEP_CB", "8,16,32").split(",")]
splits = [int(x) for x in os.environ.get("SWEEP_SPLITS", "3,8,16").split(",")]
ppls = [int(x) for x in os.environ.get("SWEEP_PPL", "4").split(",")]
seeds = [int(x) for x in os.environ.get("SWEEP_SEED", "1").split(",")]
print("n=%d m=%d iters=%d" % (n, m, iters), flush=True)
for var, ppl, cb, sp, seed in itertools.product(variants, ppls, cbs, splits, seeds):
    os.environ.update(ICP_AMD_SEARCH=var, ICP_AMD_PPL=str(ppl), ICP_AMD_CB=str(cb), ICP_AMD_QSPLITS=str(sp), ICP_AMD_SEED=str(seed))
    e = icp_amd.Engine(kernel_timing=True)
    e.set_target(Q)
    e.set_source(P)
    e.iterate(1)            # first iteration is unseeded
    e.sync()
    first_ms = e.kernel_ms()[0]
    ms = e.iterate_timed(iters) / iters
    k_ms = e.kernel_ms()[0]
    g = e.search_geometry()
    e.close()
    if g["variant"] != var:
        print("(requested %s ppl=%d, engine chose %s)" % (var, ppl, g["variant"]))
    frac = 8.0 * n * m / (k_ms * 1e-3) / 39.3e12
    print("%-5s ppl=%d cb=%2d splits=%2d seed=%d wgs=%5d | iter %.3f ms  search %.3f ms (first, unseeded %.3f)  frac_of_39.3TF %.3f  evals/s %.3e"
          % (var, ppl, cb, g["q_splits"], seed, g["workgroups"], ms, k_ms, first_ms, frac, n * float(m) / (ms * 1e-3)), flush=True)
