#!/usr/bin/env python3
"""Per-iteration time of the first iterations (un-seeded pass, then shrinking search radii) on the 1M sheet and cube."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import icp_amd
from icp_amd import synth
from percell_sweep import sheet
for name in ("sheet", "cube"):
    P, Q = sheet(1000000) if name == "sheet" else synth.make_pair(1000000, 1000000, seed=5)
    for pc in ("1", "2", "4"):
        os.environ["ICP_AMD_GRID_PER_CELL"] = pc
        e = icp_amd.Engine(search="grid")
        t0 = time.perf_counter(); e.set_target(Q); t1 = time.perf_counter(); e.set_source(P); t2 = time.perf_counter()
        its = [e.iterate_timed(1) for _ in range(8)]
        st = e.grid_stats()
        e.close()
        print("%-5s per_cell %s: set_target %.2f ms, set_source %.2f ms, iterations (ms): %s  %s" % (
            name, pc, (t1 - t0) * 1e3, (t2 - t1) * 1e3, " ".join("%.3f" % x for x in its), st), flush=True)
