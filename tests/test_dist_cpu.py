"""CPU, world_size 2 over gloo: the P-sharding protocol of icp_amd.dist (see tests/dist_worker.py)."""
import os
import subprocess
import sys

import numpy as np

import icp_amd
from conftest import ROOT
from icp_amd import dist as icpdist


def test_shard_bounds_partition():
    for n, w in ((1000000, 8), (1000000, 4), (5000, 2), (1500, 4), (1, 8), (1024, 2), (1025, 2)):
        seen = 0
        for r in range(w):
            lo, hi = icpdist.shard_bounds(n, r, w)
            assert lo == min(n, r * icp_amd.rows_per_rank(n, w)) and lo <= hi <= n
            assert lo % 1024 == 0 or lo == n
            assert lo == seen or lo == hi == n
            seen = max(seen, hi)
        assert seen == n
    # at N = 1M over 8 GPUs the largest shard is within 1% of the ideal
    assert icp_amd.rows_per_rank(1000000, 8) / (1000000 / 8) < 1.01


def test_gloo_world2_sharded_iteration():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", WORLD_SIZE="2", OMP_NUM_THREADS="2")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py")],
                              env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(2)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, o)
        assert "rank %d ok" % r in o
