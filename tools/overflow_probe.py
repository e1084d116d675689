#!/usr/bin/env python3
"""How many rows the grid search hands to its overflow kernel, iteration by iteration (1M x 1M registration)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import icp_amd
from icp_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
P, Q = synth.make_pair(n, n)
e = icp_amd.Engine(search="grid")
e.set_target(Q); e.set_source(P)
out = []
for it in range(120):
    e.iterate(1); e.sync()
    out.append(e.grid_stats()["last_overflow"])
e.close()
print("rows queued for the overflow kernel per iteration:", out)
