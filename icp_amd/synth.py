"""Deterministic synthetic point clouds for the benchmark and the parity tests.

SplitMix64 -> u = (x >> 11) * 2^-53.  Q is uniform in the cube [-1,1)^3; P is a permuted, rotated,
translated, noised copy of Q (rotation 0.10 rad about (1,2,3)/sqrt(14), t0 = (0.05,-0.03,0.02),
sigma = 1e-3), so ICP has a real alignment to find.  The reference ships no generator; this one is new.
"""
import numpy as np

_GOLD = np.uint64(0x9E3779B97F4A7C15)
SEED_Q = 20250117


def splitmix64(seed, n):
    """First n outputs of SplitMix64 seeded with `seed` (vectorised; uint64 arithmetic wraps)."""
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + _GOLD * np.arange(1, n + 1, dtype=np.uint64)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def uniform01(seed, n):
    return (splitmix64(seed, n) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def rotation(angle, axis=(1.0, 2.0, 3.0)):
    a = np.asarray(axis, dtype=np.float64)
    a = a / np.sqrt((a * a).sum())
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) + np.sin(angle) * K + (1 - np.cos(angle)) * (K @ K)


def make_pair(n, m=None, seed=SEED_Q, angle=0.10, t0=(0.05, -0.03, 0.02), sigma=1e-3):
    """Returns (P[n,3], Q[m,3]) float64, C-contiguous.  P is built from the first n rows of a
    pseudo-random permutation of Q (cycled when n > m)."""
    m = n if m is None else m
    Q = 2.0 * uniform01(seed, 3 * m).reshape(m, 3) - 1.0
    perm = np.argsort(splitmix64(seed + 1, m), kind="stable")[np.arange(n) % m]
    u = uniform01(seed + 2, 6 * n).reshape(n, 6)
    g = np.sqrt(-2.0 * np.log(1.0 - u[:, :3])) * np.cos(2.0 * np.pi * u[:, 3:])
    P = Q[perm] @ rotation(angle).T + np.asarray(t0) + sigma * g
    return np.ascontiguousarray(P), np.ascontiguousarray(Q)


def make_batch(batch, n, seed=SEED_Q):
    """`batch` independent registrations of n points each: (P[batch,n,3], Q[batch,n,3])."""
    Ps, Qs = [], []
    for b in range(batch):
        P, Q = make_pair(n, n, seed=seed + 3 * b + 3, angle=0.05 + 0.0001 * b)
        Ps.append(P)
        Qs.append(Q)
    return np.stack(Ps), np.stack(Qs)
