#!/usr/bin/env python3
"""Targets per grid cell: iteration time at 100k and 1M (volume-filling synthetic pair and a sheet)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import icp_amd
from icp_amd import synth
def sheet(n):
    rng = np.random.RandomState(1)
    u = rng.uniform(-1, 1, (n, 2))
    Q = np.column_stack([u[:, 0], u[:, 1], 0.1 * np.sin(5 * u[:, 0]) * np.cos(4 * u[:, 1])])
    c, s = np.cos(0.05), np.sin(0.05)
    P = Q[rng.permutation(n)] @ np.array([[c, -s, 0], [s, c, 0], [0, 0, 1.0]]).T + np.array([0.01, -0.02, 0.015]) + 1e-4 * rng.standard_normal((n, 3))
    return P, Q
def main():
  for name, n in (("cube", 100000), ("cube", 1000000), ("sheet", 1000000)):
      P, Q = synth.make_pair(n, n, seed=5) if name == "cube" else sheet(n)
      line = []
      for pc in ("1", "1.5", "2", "3", "4", "6", "8"):
          os.environ["ICP_AMD_GRID_PER_CELL"] = pc
          e = icp_amd.Engine(search="grid")
          e.set_target(Q); e.set_source(P)
          e.iterate(6); e.sync()
          us = min(e.iterate_timed(40) for _ in range(3)) / 40 * 1e3
          e.close()
          line.append("%s: %.1f" % (pc, us))
      print("%-5s %7d  us/iteration by targets per cell  %s" % (name, n, " | ".join(line)), flush=True)

if __name__ == "__main__":
    main()
