#!/usr/bin/env python3
"""Brute force vs grid on tiny clouds: where should AUTO switch?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import icp_amd
from icp_amd import synth
for n in (30, 100, 256, 512, 1000, 2000):
    P, Q = synth.make_pair(n, n, seed=5)
    out = []
    for search in ("f32", "grid"):
        e = icp_amd.Engine(search=search)
        e.set_target(Q); e.set_source(P)
        e.iterate(5); e.sync()
        out.append(min(e.iterate_timed(100) for _ in range(3)) / 100 * 1e3)
        e.close()
    print("n=m=%5d: f32 %.1f us, grid %.1f us per iteration" % (n, out[0], out[1]), flush=True)
