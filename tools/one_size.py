#!/usr/bin/env python3
"""Run `iters` engine iterations on an n x m synthetic pair (for rocprofv3 timelines of one size)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import icp_amd
from icp_amd import synth
n = int(sys.argv[1]); m = int(sys.argv[2]); iters = int(sys.argv[3]) if len(sys.argv) > 3 else 100
P, Q = synth.make_pair(n, m, seed=5)
e = icp_amd.Engine()
e.set_target(Q); e.set_source(P)
e.iterate(iters); e.sync()
print(e.search_geometry())
e.close()
