#!/usr/bin/env python3
"""Seeded brute-force iterations at one size with whatever libicp_amd.so is in place: search-kernel time per launch and a
checksum of the correspondences (A/B runs of kernel builds must print the same checksum)."""
import os, sys, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import icp_amd
from icp_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
m = int(sys.argv[2]) if len(sys.argv) > 2 else n
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 8
P, Q = synth.make_pair(n, m)
e = icp_amd.Engine(search=os.environ.get("AB_SEARCH", "f32"), kernel_timing=True)
e.set_target(Q); e.set_source(P)
e.iterate(1); e.sync()
first = e.kernel_ms()[0]
e.iterate(2); e.sync(); e.kernel_ms()
ms = e.iterate_timed(iters) / iters
k = e.kernel_ms()
last = e.get_last()
print("%s  n=%d m=%d  first launch %.3f ms  seeded search kernel %.3f ms avg over %d  iteration %.3f ms  idx crc %08x  rms %.17g  %s"
      % (sys.argv[4] if len(sys.argv) > 4 else "", n, m, first, k[0], k[2], ms, zlib.crc32(last["idx"].tobytes()), last["rms"], e.search_geometry()), flush=True)
e.close()
