"""First (unseeded) brute-force iteration against the seeded ones: python tools/first_iter_probe.py [N] [M]."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import icp_amd
from icp_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
m = int(sys.argv[2]) if len(sys.argv) > 2 else n
P, Q = synth.make_pair(n, m)
for search in ("f32",):
    for rep in range(2):
        e = icp_amd.Engine(search=search)
        e.set_target(Q)
        e.set_source(P)
        e.synchronize() if hasattr(e, "synchronize") else None
        t0 = time.perf_counter(); e.iterate(1); e.get_last(); t1 = time.perf_counter()
        e.iterate(5); last = e.get_last(); t2 = time.perf_counter()
        idx = last["idx"].astype(np.int64)
        print("%s %d x %d  first %.2f ms  then %.2f ms/iteration  idx hash %d" % (search, n, m, (t1 - t0) * 1e3, (t2 - t1) / 5 * 1e3,
              int((idx * (np.arange(idx.size) % 1009 + 1)).sum())), flush=True)
        e.close()
