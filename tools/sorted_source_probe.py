#!/usr/bin/env python3
"""What a spatially sorted SOURCE STORAGE order would be worth: the 1M x 1M registration with the source rows sorted by a
coarse cell on the host beforehand (numpy), the engine's own row permutation switched off (ICP_AMD_GRID_ORDER=0), with the
moving-cloud kernel (default) or the fused storage-order kernel throughout (ICP_AMD_GRID_ORDER_SWITCH=1e30).
Usage: sorted_source_probe.py [n] [per_cell]   (environment decides the variant; prints ms per iteration by tens)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import icp_amd
from icp_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
per_cell = float(sys.argv[2]) if len(sys.argv) > 2 else 128.0
P, Q = synth.make_pair(n, n)
if os.environ.get("PROBE_SORT", "1") != "0":
    lo, hi = P.min(0), P.max(0)
    dim = max(1, int(round((n / per_cell) ** (1.0 / 3.0))))
    c = np.minimum(((P - lo) / (hi - lo) * dim).astype(np.int64), dim - 1)
    key = (c[:, 2] * dim + c[:, 1]) * dim + c[:, 0]
    P = np.ascontiguousarray(P[np.argsort(key, kind="stable")])
e = icp_amd.Engine(search="grid")
e.set_target(Q); e.set_source(P)
e.iterate(3); e.sync()
ms = [e.iterate_timed(10) / 10 for _ in range(16)]
e.close()
print("sort=%s order=%s switch=%s: %s  total %.2f ms" % (os.environ.get("PROBE_SORT", "1"), os.environ.get("ICP_AMD_GRID_ORDER", "1"),
      os.environ.get("ICP_AMD_GRID_ORDER_SWITCH", "default"), " ".join("%.4f" % m for m in ms), 10 * sum(ms)), flush=True)
