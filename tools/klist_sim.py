"""CPU estimate (numpy + scipy's k-d tree; no product code) of how many rows of the synthetic pair would have to walk the
grid per iteration if a row kept K nearest targets with their certificates instead of one.  Usage: klist_sim.py [n] [iters]
Prints, per iteration, the fraction of rows that walk for K = 1, 2, 3, 4 and the fraction whose match really changed.
BETA=b: smallest search radius in cell edges (stronger box bound); NOB=1: the box bound ignored (what K alone could give)."""
import os
import sys

import numpy as np
from scipy.spatial import cKDTree

sys.path.insert(0, ".")
from icp_amd.synth import make_pair  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 40
KMAX = 4
BETA = float(os.environ.get("BETA", "0"))   # smallest search radius in cell edges
CELLS = []
P, Q = make_pair(n)
Qc = Q.mean(0)
Qz = Q - Qc
tree = cKDTree(Qz)
h = (2.0 * 8.0 / n) ** (1.0 / 3.0)   # two targets per cell of the cube [-1, 1)^3
lo = Qz.min(0)


def box_bound(a, r):
    """distance from a to the nearest outer edge of the cells the box [a - r, a + r] touches"""
    r = np.maximum(r, BETA * h)
    c0 = np.floor((a - r[:, None] - lo) / h)
    c1 = np.floor((a + r[:, None] - lo) / h)
    e0 = lo + c0 * h
    e1 = lo + (c1 + 1) * h
    CELLS.append(float((c1 - c0 + 1).prod(1).mean()))
    if os.environ.get("NOB"):
        return np.full(len(a), np.inf)
    return np.minimum((a - e0).min(1), (e1 - a).min(1))


state = {K: dict(ids=np.full((n, K), -1), cert=np.zeros((n, K))) for K in range(1, KMAX + 1)}
prev = None
prev_nn = None
for it in range(iters):
    Pc = P.mean(0)
    A = P - Pc
    d, ids = tree.query(A, k=KMAX + 1, workers=8)
    line = []
    delta = np.zeros(n) if prev is None else np.sqrt(((A - prev) ** 2).sum(1)) * (1 + 1e-9) + 1e-12
    for K in range(1, KMAX + 1):
        st = state[K]
        if it == 0:
            walk = np.ones(n, bool)
        else:
            st["cert"] = st["cert"] - delta[:, None]
            cand = st["ids"]
            dc = np.sqrt(((A[:, None, :] - Qz[np.maximum(cand, 0)]) ** 2).sum(2))
            dc[cand < 0] = np.inf
            # smallest k such that the best of the first k entries is certified against everything outside them
            walk = np.ones(n, bool)
            best = np.full(n, np.inf)
            newcert = st["cert"].copy()
            for k in range(K):
                best = np.minimum(best, dc[:, k])
                ok = walk & (st["cert"][:, k] > best) & (st["cert"][:, k] > 0)
                walk &= ~ok
            # rows that hold: reorder so that the winner comes first, certificates = min(own distance bound, next)
            hold = ~walk
            if hold.any():
                order = np.argsort(dc[hold], axis=1, kind="stable")
                ci = np.take_along_axis(cand[hold], order, 1)
                dd = np.take_along_axis(dc[hold], order, 1)
                oc = st["cert"][hold]
                # every target outside the first k sorted entries is at least min(dd[k], cert of the full list ...): be
                # conservative: outside the list >= oc[:, K - 1]; inside: the next sorted distance
                nc = np.empty_like(oc)
                for k in range(K):
                    nxt = dd[:, k + 1] if k + 1 < K else np.full(len(dd), np.inf)
                    nc[:, k] = np.minimum(nxt, oc[:, K - 1])
                st["ids"][hold] = ci
                st["cert"][hold] = nc
        w = walk
        r = d[w, 0] if it == 0 else np.sqrt(((A[w] - Qz[np.maximum(st["ids"][w, 0], 0)]) ** 2).sum(1))
        if it == 0:
            r = d[w, 0] * 4
        B = box_bound(A[w], r)
        st["ids"][w] = ids[w, :K]
        for k in range(K):
            st["cert"][w, k] = np.minimum(d[w, k + 1], B)
        # sanity: the held rows have the true match
        assert (st["ids"][:, 0] == ids[:, 0]).mean() > 0.999999, (K, (st["ids"][:, 0] != ids[:, 0]).sum())
        line.append(w.mean())
    changed = 1.0 if prev_nn is None else (ids[:, 0] != prev_nn).mean()
    print(f"it {it:3d}  walk K=1..{KMAX}: " + " ".join(f"{x:.3f}" for x in line) + f"   match changed {changed:.3f}   mean step {delta.mean():.2e}", flush=True)
    prev_nn = ids[:, 0].copy()
    prev = A
    # ICP update
    Qm = Qz[ids[:, 0]]
    cov = Qm.T @ A
    U, S, VT = np.linalg.svd(cov)
    R = U @ VT
    t = Qc - Pc @ R.T
    P = P @ R.T + t
