import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(autouse=True)
def no_sticky_hip_error_left_behind(request):
    """After every GPU test: the host thread's sticky HIP error word (hipPeekAtLastError) is clean.  The library reports its
    launches through their own return codes and clears nothing on behalf of others, so anything it left there would
    surface later as somebody else's failure (round 2: torch raised 'invalid device ordinal' for RCCL's leftover)."""
    yield
    if request.node.get_closest_marker("gpu") is None:
        return
    import ctypes
    try:
        hip = ctypes.CDLL("libamdhip64.so")
    except OSError:
        return
    err = int(hip.hipPeekAtLastError())
    if err != 0:
        hip.hipGetLastError()          # do not let one leak fail every later test as well
        pytest.fail("%s left sticky HIP error %d behind" % (request.node.nodeid, err), pytrace=False)


def golden(name):
    """A committed fixture (tests/golden/*.npz, vectors from the unmodified reference).  A missing file is an ERROR, never a
    skip: the fixtures are part of the tree, and a parity suite that quietly skips itself would still report success."""
    path = os.path.join(GOLDEN, name + ".npz")
    if not os.path.exists(path):
        pytest.fail("golden fixture %s is missing from tests/golden/ (it is committed with the tree: check .gpurunignore / "
                    "the checkout)" % name, pytrace=False)
    with np.load(path, allow_pickle=False) as z:
        g = {k: z[k] for k in z.files}
    if "P" in g and "P_in_iters" in g and len(g["P_in_iters"]) and g["P_in_iters"][0] == 0:
        # iteration 0 starts from P itself; make_golden.py does not store it twice
        g["P_in"] = np.concatenate([g["P"][None], g["P_in"].reshape(-1, *g["P"].shape)])
    return g


def stored_clouds(g, shape):
    """[(iteration, running cloud the reference held at its start)] for the stored iterations after the first of an x20 fixture
    (the first one starts from the input cloud itself)."""
    its = [int(k) for k in g["P_in_iters"]]
    P_in = np.asarray(g["P_in"])
    later = [k for k in its if k > 0]
    P_in = P_in.reshape(-1, *shape)
    if len(P_in) == len(its):          # golden() put the input cloud back in front
        P_in = P_in[len(its) - len(later):]
    assert len(P_in) == len(later)
    return list(zip(later, P_in))


ICP_CASES = ["c1_30pts", "cow_tr1", "syn256", "syn1k", "syn3k_4k", "lattice", "syn20k", "horse_tr1", "bun045_000"]
SMALL_ICP_CASES = ["c1_30pts", "cow_tr1", "syn256", "syn1k", "syn3k_4k", "lattice"]


def relerr(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    scale = max(np.abs(b).max(), 1e-300)
    return float(np.abs(a - b).max() / scale)


def relerr_t(t, t_ref, Qc, Pc):
    """t = Qc - Pc.R^T cancels to ~0 once the clouds are aligned, so its error is measured against the
    size of the operands it is computed from (max of |t|, |Qc|, |Pc|), not against |t| alone."""
    scale = max(np.abs(t_ref).max(), np.abs(Qc).max(), np.abs(Pc).max(), 1e-300)
    return float(np.abs(np.asarray(t) - np.asarray(t_ref)).max() / scale)
