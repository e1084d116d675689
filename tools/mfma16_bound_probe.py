#!/usr/bin/env python3
"""Worst |filter value - exact| / E' of the fp16 MFMA pre-filter (icp_amd_test_mfma16_bound) over a few kinds of data."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import icp_amd
rng = np.random.RandomState(12)
def probe(P, Qz):
    P, Qz = np.ascontiguousarray(P), np.ascontiguousarray(np.repeat(Qz, 32, axis=0))
    w = C.c_double()
    icp_amd.check(icp_amd.lib().icp_amd_test_mfma16_bound(P.ctypes.data_as(C.c_void_p), C.c_size_t(len(P)), Qz.ctypes.data_as(C.c_void_p),
                                                          C.c_size_t(len(Qz)), C.byref(w)))
    return w.value
q, p = rng.uniform(-1, 1, (512, 3)), rng.uniform(-1, 1, (4000, 3))
print("uniform cube:                      %.4f" % probe(p, q))
print("scaled by 1e-9 / 1e+9:             %.4f / %.4f" % (probe(p * 1e-9, q * 1e-9), probe(p * 1e9, q * 1e9)))
q2 = q.copy(); q2[:256] *= 10.0 ** rng.uniform(-9, -2, (256, 1)); p2 = p.copy(); p2[:2000] *= 10.0 ** rng.uniform(-9, -2, (2000, 1))
print("coordinates tiny against qmax:     %.4f" % probe(p2, q2))
p3 = p.copy(); p3[::2] *= 60.0
print("sources up to 60 x the extent:     %.4f" % probe(p3, q))
q4 = q.copy(); q4[:, 2] = 0; p4 = p.copy(); p4[:, 0] = 0
print("flat clouds (exact zeros):         %.4f" % probe(p4, q4))
lat = rng.randint(-8, 9, (512, 3)).astype(float)
print("integer lattice:                   %.4f" % probe(rng.randint(-8, 9, (4000, 3)) + 0.5, lat))
