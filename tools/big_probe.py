#!/usr/bin/env python3
"""Beyond the benchmark sizes: grid vs brute force at N = M = 4M (bit-identical results expected), grid alone at 16M."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import icp_amd
from icp_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000000
P, Q = synth.make_pair(n, n, seed=9)
outs = {}
for search in ("grid", "f32") if n <= 4000000 else ("grid",):
    e = icp_amd.Engine(search=search)
    e.set_target(Q); e.set_source(P)
    e.iterate(2); e.sync()
    ms = e.iterate_timed(3) / 3
    outs[search] = (e.get_last()["idx"], e.get_last()["cov"])
    print("%d x %d %-4s: %.3f ms per iteration, rms %.3e" % (n, n, search, ms, e.get_last()["rms"]), flush=True)
    e.close()
if len(outs) == 2:
    same = np.array_equal(outs["grid"][0], outs["f32"][0]) and np.array_equal(outs["grid"][1], outs["f32"][1])
    print("grid == brute force (idx, cov bit for bit):", same, flush=True)
    sys.exit(0 if same else 1)
