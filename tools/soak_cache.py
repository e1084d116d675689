#!/usr/bin/env python3
"""Soak of the device-memory cache (devmem.cpp): a few hundred one-shot registrations of random sizes through the one-shot,
engine, batch and stage-level entry points.  Checks: a repeated input gives the bits it gave the first time (recycled, dirty
blocks change nothing), the memory the driver reports in use stays bounded (no leak), and the cache stays under its cap."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import icp_amd
from icp_amd import synth

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.RandomState(9)
first = {}
free0, total = torch.cuda.mem_get_info(0)
peak_used = 0
base_used = 0
t_say = time.time() + 20
for k in range(rounds):
    kind = rng.randint(4)
    n, m = int(rng.choice([300, 1000, 2903, 10000, 48485, 120000, 300000])), int(rng.choice([500, 1000, 2903, 10000, 48485, 120000, 300000]))
    seed = int(rng.randint(5))
    key = (kind, n, m, seed)
    P, Q = synth.make_pair(n, m, seed=100 + seed)
    if kind == 0:
        out = icp_amd.icp_gpu_optimized(P, Q, 3)
    elif kind == 1:
        e = icp_amd.Engine(search=["auto", "f32", "grid"][seed % 3])
        e.set_target(Q); e.set_source(P); e.iterate(2)
        e.set_target(Q[: max(1, m // 2)]); e.iterate(1)          # a target of another size under the same source
        out = e.get_source(); e.close()
    elif kind == 2:
        b, nn = 1 + seed * 7, min(n, 900)
        Pb, Qb = synth.make_batch(b, nn)
        out = icp_amd.icp_gpu_batch(Pb, Qb, 3)["P_out"]
        key = (kind, b, nn, 0)
    else:
        Pc = P.mean(0)
        out = icp_amd.correspondences(P[: min(n, 20000)], Pc, Q - Q.mean(0), search="f32", q_splits=1 + seed)
    out = np.asarray(out)
    if key in first:
        assert np.array_equal(first[key], out), ("result changed on a recycled block", key)
    elif len(first) < 200:
        first[key] = out.copy()
    free, _ = torch.cuda.mem_get_info(0)
    peak_used = max(peak_used, free0 - free)
    if k == 59:   # the runtime's own pools (code objects, staging buffers of pageable copies, scratch) are up by now
        icp_amd.cache_trim()
        base_used = free0 - torch.cuda.mem_get_info(0)[0]
        print("baseline after 60 registrations and a trim: %.0f MiB in use beyond the start" % (base_used / 2**20), flush=True)
    if time.time() > t_say or k == rounds - 1:
        print("... %d registrations, %d distinct inputs, device memory in use beyond the start: %.0f MiB (peak %.0f)" % (
            k + 1, len(first), (free0 - free) / 2**20, peak_used / 2**20), flush=True)
        t_say = time.time() + 20
held = icp_amd.cache_trim()
free, _ = torch.cuda.mem_get_info(0)
print("cache held %.0f MiB at the end; after trim the driver reports %.0f MiB in use beyond the start" % (held / 2**20, (free0 - free) / 2**20))
assert held <= 2048 * 2**20
assert (free0 - free) - base_used < 32 * 2**20, "memory not returned: the library leaks device memory"
print("soak ok")
