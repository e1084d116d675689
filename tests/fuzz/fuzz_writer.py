#!/usr/bin/env python3
"""Differential check of the result writer and the final report (icp_amd_write_cloud, icp_amd_euclidian_distance) against
the reference executable itself (oracle/_ref/CPUICP = the reference's src/cpu/main.cpp, unmodified) -- CPU only, build
container only.  CPUICP registers a random pair and writes `point_cloud_result` + its "Squared actual mean diff" line;
ref_trace gives the very doubles it wrote (same code, same inputs); the library's writer must reproduce the file byte
for byte from those doubles, and its report the printed value, across scales 1e-8 .. 1e8 (both ostream notations).
    python tests/fuzz/fuzz_writer.py [cases] [seed]"""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np
import icp_amd
from make_golden import TRACE, write_csv, read_trace
CPUICP = os.path.join(ROOT, "oracle", "_ref", "CPUICP")


def run(cases, seed, verbose=True):
    rng = np.random.RandomState(seed)
    bad, compared = [], 0
    for k in range(cases):
        n = int(rng.randint(1, 120))
        s = 10.0 ** rng.uniform(-8, 8)
        Q = rng.uniform(-1, 1, (n, 3)) * s + rng.choice([0.0, 1.0]) * s * 10.0 ** rng.uniform(-2, 3)
        P = Q[rng.permutation(n)] + rng.standard_normal((n, 3)) * s * 10.0 ** rng.uniform(-9, -1)
        if rng.rand() < 0.2: P[:, 2] = 0.0; Q[:, 2] = 0.0           # exact zeros and a rank-deficient covariance
        iters = int(rng.randint(1, 4))
        with tempfile.TemporaryDirectory() as td:
            pf, qf, tb = os.path.join(td, "P.csv"), os.path.join(td, "Q.csv"), os.path.join(td, "t.bin")
            write_csv(pf, P); write_csv(qf, Q)
            r = subprocess.run([CPUICP, pf, qf, str(iters)], cwd=td, capture_output=True, text=True)
            if r.returncode != 0: continue
            subprocess.check_call([TRACE, pf, qf, str(iters), tb])
            P_out = read_trace(tb)["P_out"]
            want_file = open(os.path.join(td, "point_cloud_result"), "rb").read()
            mine = os.path.join(td, "mine")
            icp_amd.write_cloud(mine, P_out)
            got_file = open(mine, "rb").read()
            want_line = [l for l in r.stdout.splitlines() if l.startswith("Squared actual mean diff")][-1]
            got_line = "Squared actual mean diff: " + ("%g" % icp_amd.euclidian_distance(Q, P_out))   # std::cout << double, as GPUICP prints it
            compared += 1
            why = "file" if want_file != got_file else "report %r vs %r" % (want_line, got_line) if want_line != got_line else None
            if why:
                bad.append((k, why))
                if verbose: print("MISMATCH case %d (n=%d scale %.3g): %s" % (k, n, s, why), flush=True)
    return compared, bad


if __name__ == "__main__":
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    compared, bad = run(cases, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    print("%d cases, %d compared (the reference ran), %d mismatches" % (cases, compared, len(bad)))
    sys.exit(1 if bad else 0)
