#!/usr/bin/env python3
"""Fuzz of the P-sharded path on one GPU (W engines, all-gathers emulated by device copies, as in
tests/test_gpu_sharded.py): random sizes and world sizes, including ranks that own no rows; results must be
bit-identical to the single-engine run, on the AUTO (grid) and the brute-force search."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import icp_amd
from icp_amd import synth
from test_gpu_sharded import run_sharded

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t_end = time.time() + budget
cases = bad = 0
t_say = time.time() + 30
while time.time() < t_end:
    if time.time() > t_say:
        print("... %d cases, %d mismatches so far" % (cases, bad), flush=True)
        t_say = time.time() + 30
    n, m, world, iters = int(rng.randint(1, 40000)), int(rng.randint(1, 30000)), int(rng.randint(2, 9)), int(rng.randint(1, 4))
    P, Q = synth.make_pair(n, max(n, m), seed=int(rng.randint(1 << 30)))
    Q = Q[:m] if m < len(Q) else Q
    for search in ("auto", "f32"):
        os.environ["ICP_AMD_SEARCH"] = search
        e = icp_amd.Engine()
        e.set_target(Q); e.set_source(P); e.iterate(iters)
        single, P_single = e.get_last(), e.get_source()
        e.close()
        lasts, P_out = run_sharded(P, Q, world, iters)
        ok = np.array_equal(np.concatenate([l["idx"] for l in lasts]), single["idx"]) and np.array_equal(P_out, P_single)
        ok = ok and all(np.array_equal(l[k], single[k]) for l in lasts for k in ("cov", "R", "t", "Pc"))
        if not ok:
            bad += 1
            print("MISMATCH n=%d m=%d world=%d iters=%d search=%s" % (n, len(Q), world, iters, search), flush=True)
    cases += 1
print("cases %d, mismatches %d" % (cases, bad), flush=True)
sys.exit(1 if bad else 0)
