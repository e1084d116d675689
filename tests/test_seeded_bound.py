"""CPU: the error bound behind the seeded search of the multi-registration kernel (icp_amd/csrc/kernels.hip,
scan_small_seeded), checked in exact rational arithmetic.

The kernel's fast path computes, for a centred source row a and a target record {-2q, |q|^2 + C},
    f = fma(az, -2qz, fma(ay, -2qy, fma(ax, -2qx, w))),   w = fl(fl(|q|^2) + C),
and claims  |f - F| <= E  with  F = |a - q|^2 - |a|^2 + C  (exact)  and  E = 2^-52 (11 |a| qmax + 5 qmax^2 + 8 C),
and that a target whose ORACLE distance (un-fused, src/cpu/icp.cpp:72-93 order) is <= bd has f <= thr(bd), also as high
words.  Every double operation is emulated here with fractions.Fraction (float(Fraction) rounds correctly), so the check
does not depend on the machine's own arithmetic.  No GPU, no oracle library."""
import math
import struct
from fractions import Fraction as Fr

import numpy as np


def fl(x):
    return float(x)


def fma(a, b, c):
    return float(Fr(a) * Fr(b) + Fr(c))


def hi_word(x):
    return struct.unpack("<Q", struct.pack("<d", x))[0] >> 32


def norm2(x, y, z):   # (x*x + y*y) + z*z, un-fused
    return fl(Fr(fl(Fr(fl(Fr(x) * Fr(x))) + Fr(fl(Fr(y) * Fr(y))))) + Fr(fl(Fr(z) * Fr(z))))


def d2_oracle(a, q):  # the oracle's distance: differences, squares, two adds, each rounded
    d = [fl(Fr(a[k]) - Fr(q[k])) for k in range(3)]
    return norm2(*d)


def offset_C(v):      # ldexp(1, frexp_exp(v) + 2): a power of two in (4 v, 8 v]
    return math.ldexp(1.0, math.frexp(v)[1] + 2)


def check_case(rng, scale, a_ratio, q_ratio, n_rows=6, n_tgt=24):
    A = rng.uniform(-1, 1, (n_rows, 3)) * scale * a_ratio
    Q = rng.uniform(-1, 1, (n_tgt, 3)) * scale * q_ratio
    Q[0] = A[0] + rng.uniform(-1, 1, 3) * scale * 1e-9          # a near coincidence
    Q[1] = A[1]                                                 # an exact one
    qn = [norm2(*q) for q in Q]
    qmax2 = max(qn)
    pmax = max(norm2(*a) for a in A)
    C = offset_C(max(pmax, qmax2))
    qmax = math.sqrt(qmax2)
    worst = 0.0
    for a in A:
        pn = norm2(*a)
        E = 2.0 ** -52 * (11.0 * math.sqrt(pn) * qmax + 5.0 * qmax2 + 8.0 * C)
        assert C > 2.0 * pn + 4.0 * E                           # the kernel's per-row test holds by construction of C
        off = fl(Fr(fl(Fr(C) - Fr(fl(Fr(pn) * Fr(1.0 - 2.0 ** -50))))) + Fr(E))
        pn_exact = sum(Fr(v) * Fr(v) for v in a)
        for q, w0 in zip(Q, qn):
            w = fl(Fr(w0) + Fr(C))
            f = fma(a[2], -2.0 * q[2], fma(a[1], -2.0 * q[1], fma(a[0], -2.0 * q[0], w)))
            F = sum((Fr(a[k]) - Fr(q[k])) ** 2 for k in range(3)) - pn_exact + Fr(C)
            err = abs(Fr(f) - F)
            assert err <= Fr(E), (scale, a_ratio, q_ratio, float(err), E)
            assert f > 0.0
            worst = max(worst, float(err / Fr(E)))
            # the threshold at the tightest bound this target must pass: bd = its own oracle distance
            bd = d2_oracle(a, q)
            thr = fl(Fr(fma(bd, 1.0 + 2.0 ** -50, off)) * Fr(1.0 + 2.0 ** -50))
            assert f <= thr, (scale, a_ratio, q_ratio, f, thr)
            assert hi_word(f) <= hi_word(thr)
    return worst


def test_seeded_form_error_bound_and_threshold_hold_in_exact_arithmetic():
    rng = np.random.RandomState(5)
    worst = 0.0
    for scale in (1.0, 1e-3, 7.0, 1e5, 1e-60, 1e60, 1e-130, 1e130):
        for a_ratio, q_ratio in ((1.0, 1.0), (1.0, 1e-3), (1e-3, 1.0), (1.0, 0.4), (0.05, 1.0), (1.0, 1e-9)):
            for _ in range(3):
                worst = max(worst, check_case(rng, scale, a_ratio, q_ratio))
    assert 0.0 < worst < 1.0   # and the bound is not vacuous: the largest observed error is a visible fraction of it
    print("largest |f - F| / E observed: %.3f" % worst)


def test_expanded_form_error_bound_holds_in_exact_arithmetic():
    """The first iteration's filter (scan_small_expanded): f = fma chain on {-2q, |q|^2}, F = |a - q|^2 - |a|^2,
    |f - F| <= E = 2^-52 (11 |a| qmax + 5 qmax^2)."""
    rng = np.random.RandomState(6)
    worst = 0.0
    for scale in (1.0, 1e-4, 1e7, 1e-90, 1e90):
        for a_ratio, q_ratio in ((1.0, 1.0), (1.0, 1e-3), (1e-3, 1.0), (30.0, 1.0), (1.0, 30.0)):
            A = rng.uniform(-1, 1, (8, 3)) * scale * a_ratio
            Q = rng.uniform(-1, 1, (40, 3)) * scale * q_ratio
            Q[0] = A[0]
            qn = [norm2(*q) for q in Q]
            qmax2 = max(qn)
            qmax = math.sqrt(qmax2)
            for a in A:
                pn = norm2(*a)
                E = 2.0 ** -52 * (11.0 * math.sqrt(pn) * qmax + 5.0 * qmax2)
                pn_exact = sum(Fr(v) * Fr(v) for v in a)
                for q, w in zip(Q, qn):
                    f = fma(a[2], -2.0 * q[2], fma(a[1], -2.0 * q[1], fma(a[0], -2.0 * q[0], w)))
                    F = sum((Fr(a[k]) - Fr(q[k])) ** 2 for k in range(3)) - pn_exact
                    err = abs(Fr(f) - F)
                    assert err <= Fr(E), (scale, a_ratio, q_ratio, float(err), E)
                    worst = max(worst, float(err / Fr(E)))
    assert 0.0 < worst < 1.0
    print("expanded form: largest |f - F| / E observed: %.3f" % worst)


def test_offset_is_a_power_of_two_above_four_times_the_range():
    for v in (1e-300, 3.0, 4.0, 0.75, 1.0, 123456.789, 1e279):
        C = offset_C(v)
        assert 4.0 * v < C <= 8.0 * v
        assert math.frexp(C)[0] == 0.5
