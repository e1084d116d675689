#!/usr/bin/env python3
"""Wall time of the GPUICP executable end to end (CSV parse, registration, report, result file) at N = M = 1M, 20 iterations."""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import icp_amd
from icp_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
P, Q = synth.make_pair(n, n, seed=5)
with tempfile.TemporaryDirectory() as d:
    for name, A in (("P.txt", P), ("Q.txt", Q)):
        np.savetxt(os.path.join(d, name), A, fmt="%.17g", delimiter=",", header="Points_0,Points_1,Points_2", comments="")
        icp_amd.write_cloud_binary(os.path.join(d, name.replace(".txt", ".npy")), A)
    exe = os.path.join(ROOT, "bin", "GPUICP")
    for args in (["P.txt", "Q.txt"], ["P.npy", "Q.npy"]):
        best = 1e9
        for _ in range(2):
            t0 = time.perf_counter()
            r = subprocess.run([exe] + args + ["20", "-shared-loop"], cwd=d, capture_output=True, text=True, timeout=300)
            best = min(best, time.perf_counter() - t0)
        assert r.returncode == 0, r.stderr
        print("GPUICP %s %s 20 -shared-loop (n = m = %d): %.2f s wall, %s" % (args[0], args[1], n, best, r.stdout.strip()), flush=True)
