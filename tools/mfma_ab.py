#!/usr/bin/env python3
"""A/B of the two fp32 pre-filters of the brute-force search (matrix pipe vs packed VALU): same clouds, same results,
time per seeded iteration from the engine's own kernel timing.  Each configuration runs in a child process because
ICP_AMD_MFMA is read once per engine plan."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, json, time
sys.path.insert(0, %r)
import numpy as np, icp_amd
from icp_amd import synth
n, m, iters = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
P, Q = synth.make_pair(n, m, seed=5)
e = icp_amd.Engine(search="f32")
e.set_target(Q); e.set_source(P)
e.iterate(2); e.sync()
t0 = time.perf_counter(); e.iterate(iters); e.sync(); dt = (time.perf_counter() - t0) / iters
idx = e.get_last()["idx"]
print(json.dumps({"ms_per_iteration": dt * 1e3, "geometry": e.search_geometry(), "idx_sum": int(idx.astype(np.int64).sum()),
                  "idx_hash": int(np.bitwise_xor.reduce(idx.astype(np.int64) * (np.arange(idx.size) + 1)))}))
e.close()
''' % ROOT
def run(n, m, iters, env):
    r = subprocess.run([sys.executable, "-c", CHILD, str(n), str(m), str(iters)], env={**os.environ, **env}, capture_output=True, text=True)
    if r.returncode: raise SystemExit(r.stderr[-3000:])
    return json.loads(r.stdout.strip().splitlines()[-1])
sizes = [(int(a), int(b), int(c)) for a, b, c in (s.split("x") for s in sys.argv[1:])] or [(1000000, 1000000, 5), (125952, 1000000, 20), (100000, 100000, 50), (20000, 20000, 100)]
for n, m, it in sizes:
    for label, env in (("valu", {"ICP_AMD_MFMA": "0"}),
                       ("h16 g2", {"ICP_AMD_MFMA": "16", "ICP_AMD_MFMA_G": "2"}), ("h16 g4", {"ICP_AMD_MFMA": "16", "ICP_AMD_MFMA_G": "4"})
                       ):
        d = run(n, m, it, env)
        print(f"{n:8d} x {m:8d} {label:6s} {d['ms_per_iteration']:9.3f} ms/iteration  geometry {d['geometry']}  idx hash {d['idx_hash']}", flush=True)
