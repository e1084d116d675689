"""icp_amd -- MI355X-native Iterative Closest Point engine (host-side mirror of the reference interface).

The product is the C-ABI library icp_amd/libicp_amd.so (hand-written HIP kernels for gfx950, see
include/icp_amd.h and DESIGN.md) and the GPUICP executable built on it.  This package is the thin Python
face of that ABI used by the tests and bench.py; its function names follow the reference's C++ entry
points (reference include/gpu/icp.cuh:38-39, include/libCSV/csv.hpp:33).
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import METHODS, SEARCH, IcpAmdError, Options, Trace, check, lib

__all__ = ["icp_gpu_optimized", "icp_gpu", "icp_gpu_multi", "icp_gpu_batch", "Batch", "Engine", "read_csv", "read_csv_mt", "read_cloud", "write_cloud", "write_cloud_binary", "euclidian_distance",
           "correspondences", "cross_cov", "centroid", "solve", "apply", "rows_per_rank", "chunks_per_rank",
           "IcpAmdError", "device_count"]


def _f64(a, cols=3):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if a.ndim != 2 or a.shape[1] != cols:
        raise IcpAmdError(_lib.EINVAL, "expected an (n, %d) array, got %r" % (cols, a.shape))
    return a


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def device_count():
    return int(lib().icp_amd_device_count())


def init(device=-1):
    """icp_amd_init: create the HIP context and load the code objects now (e.g. on another thread, beside file parsing)."""
    check(lib().icp_amd_init(C.c_int(device)))


def cache_trim():
    """icp_amd_cache_trim: give the device memory the library keeps for reuse back to the driver; returns the bytes released."""
    f = lib().icp_amd_cache_trim
    f.restype = C.c_size_t
    return int(f())


def rows_per_rank(n_global, world):
    return int(lib().icp_amd_rows_per_rank(n_global, world))


def chunks_per_rank(n_global, world):
    return int(lib().icp_amd_chunks_per_rank(n_global, world))


def icp_gpu_optimized(P, Q, iterations, method="-shared-loop", trace=False):
    """Mirror of icp_gpu_optimized(CPUMatrix&, CPUMatrix&, unsigned, std::string&) (reference
    src/gpu/icp.cu:469-652).  Returns the registered cloud; with trace=True returns (P_out, dict)."""
    if method not in METHODS:
        raise IcpAmdError(_lib.EINVAL, "Invalid method, must be: -shared|-loop|-shared-loop|-batch")
    P, Q = _f64(P), _f64(Q)
    n = P.shape[0]
    out = np.empty_like(P)
    tr, d = None, None
    if trace:
        d = dict(Pc=np.empty((iterations, 3)), idx=np.empty((iterations, n), dtype=np.int32),
                 cov=np.empty((iterations, 3, 3)), S=np.empty((iterations, 3)), R=np.empty((iterations, 3, 3)),
                 t=np.empty((iterations, 3)), rms=np.empty(iterations))
        tr = Trace(**{k: v.ctypes.data for k, v in d.items()})
    check(lib().icp_amd_run(_p(P), C.c_size_t(n), _p(Q), C.c_size_t(Q.shape[0]), C.c_uint(iterations),
                            C.c_int(METHODS[method]), _p(out), C.byref(tr) if tr is not None else None))
    return (out, d) if trace else out


def icp_gpu_multi(P, Q, iterations, n_gpus, devices=None, trace=False):
    """icp_amd_run_multi: source rows sharded over n_gpus devices of this node, one process, RCCL all-gathers."""
    P, Q = _f64(P), _f64(Q)
    n = P.shape[0]
    out = np.empty_like(P)
    tr, d = None, None
    if trace:
        d = dict(Pc=np.empty((iterations, 3)), idx=np.empty((iterations, n), dtype=np.int32),
                 cov=np.empty((iterations, 3, 3)), S=np.empty((iterations, 3)), R=np.empty((iterations, 3, 3)),
                 t=np.empty((iterations, 3)), rms=np.empty(iterations))
        tr = Trace(**{k: v.ctypes.data for k, v in d.items()})
    devs = None if devices is None else (C.c_int * n_gpus)(*devices)
    check(lib().icp_amd_run_multi(_p(P), C.c_size_t(n), _p(Q), C.c_size_t(Q.shape[0]), C.c_uint(iterations), C.c_int(n_gpus),
                                  devs, _p(out), C.byref(tr) if tr is not None else None))
    return (out, d) if trace else out


def icp_gpu(P, Q, iterations, trace=False):
    """Mirror of icp_gpu(CPUMatrix&, CPUMatrix&, unsigned) (reference src/gpu/icp.cu:275-445, `-batch`)."""
    return icp_gpu_optimized(P, Q, iterations, "-batch", trace=trace)


class Engine:
    """Persistent device-resident engine (icp_amd_create ... icp_amd_destroy)."""

    def __init__(self, device=0, stream=None, search="auto", q_splits=0, kernel_timing=False, rank=0, world=1,
                 n_global=0, cov_gather=None, cen_gather=None, strict_order=False, sort_source=False):
        o = Options()
        lib().icp_amd_default_options(C.byref(o))
        o.device, o.stream, o.search, o.q_splits = device, stream, SEARCH[search], q_splits
        o.kernel_timing, o.rank, o.world, o.n_global = int(kernel_timing), rank, world, n_global
        o.cov_gather, o.cen_gather = cov_gather, cen_gather
        o.strict_order = int(strict_order)
        o.sort_source = int(sort_source)
        self._h = C.c_void_p()
        self.n = 0
        check(lib().icp_amd_create(C.byref(self._h), C.byref(o)))

    def close(self):
        if getattr(self, "_h", None):
            lib().icp_amd_destroy(self._h)
            self._h = None

    __del__ = close

    def set_target(self, Q):
        Q = _f64(Q)
        check(lib().icp_amd_set_target(self._h, _p(Q), C.c_size_t(Q.shape[0])))

    def set_source(self, P):
        P = _f64(P)
        self.n = P.shape[0]
        check(lib().icp_amd_set_source(self._h, _p(P), C.c_size_t(self.n)))

    def iterate(self, iters):
        check(lib().icp_amd_iterate(self._h, C.c_uint(iters)))

    def iterate_timed(self, iters):
        ms = C.c_double()
        check(lib().icp_amd_iterate_timed(self._h, C.c_uint(iters), C.byref(ms)))
        return ms.value

    def step_begin(self):
        check(lib().icp_amd_step_begin(self._h))

    def step_search(self):
        check(lib().icp_amd_step_search(self._h))

    def step_update(self):
        check(lib().icp_amd_step_update(self._h))

    def sync(self):
        check(lib().icp_amd_sync(self._h))

    def get_source(self):
        out = np.empty((self.n, 3))
        check(lib().icp_amd_get_source(self._h, _p(out)))
        return out

    def get_last(self):
        d = dict(Pc=np.empty(3), idx=np.empty(self.n, dtype=np.int32), cov=np.empty((3, 3)), S=np.empty(3),
                 R=np.empty((3, 3)), t=np.empty(3))
        rms = C.c_double()
        check(lib().icp_amd_get_last(self._h, _p(d["Pc"]), _p(d["idx"]), _p(d["cov"]), _p(d["S"]), _p(d["R"]),
                                     _p(d["t"]), C.byref(rms)))
        d["rms"] = rms.value
        return d

    def target_centroid(self):
        qc = np.empty(3)
        check(lib().icp_amd_get_target_centroid(self._h, _p(qc)))
        return qc

    def kernel_ms(self):
        avg, tot, cnt = C.c_double(), C.c_double(), C.c_uint()
        check(lib().icp_amd_kernel_ms(self._h, C.byref(avg), C.byref(tot), C.byref(cnt)))
        return avg.value, tot.value, cnt.value

    def set_reject_distance(self, dmax):
        check(lib().icp_amd_set_reject_distance(self._h, C.c_double(dmax)))

    def iterate_until(self, max_iters, rot_tol, trans_tol, check_every=1):
        done = C.c_uint()
        check(lib().icp_amd_iterate_until(self._h, C.c_uint(max_iters), C.c_double(rot_tol), C.c_double(trans_tol),
                                          C.c_uint(check_every), C.byref(done)))
        return done.value

    def last_motion(self):
        a, b, c = C.c_double(), C.c_double(), C.c_double()
        check(lib().icp_amd_get_last_motion(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return dict(pairs_used=a.value, delta_rot=b.value, delta_trans=c.value)

    def grid_stats(self):
        a, b, c = C.c_uint(), C.c_uint(), C.c_uint()
        check(lib().icp_amd_grid_stats(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return dict(grid_launches=a.value, brute_force_launches=b.value, last_overflow=c.value)

    def search_geometry(self):
        a, b, c, d = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        check(lib().icp_amd_search_geometry(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(d)))
        return dict(points_per_lane=a.value, q_splits=b.value, workgroups=c.value,
                    variant={1: "exact", 2: "fma", 3: "f32", 4: "grid"}.get(d.value, str(d.value)))


class Batch:
    """Multi-registration mode: `batch` independent (P_b, Q_b) pairs, one workgroup each (icp_amd_batch_*)."""

    def __init__(self, batch, n, m, device=0):
        self.batch, self.n, self.m = batch, n, m
        self._h = C.c_void_p()
        check(lib().icp_amd_batch_create(C.byref(self._h), C.c_int(device), C.c_size_t(batch), C.c_size_t(n), C.c_size_t(m)))

    def close(self):
        if getattr(self, "_h", None):
            lib().icp_amd_batch_destroy(self._h)
            self._h = None

    __del__ = close

    def upload(self, P, Q):
        P = np.ascontiguousarray(P, dtype=np.float64)
        Q = np.ascontiguousarray(Q, dtype=np.float64)
        if P.shape != (self.batch, self.n, 3) or Q.shape != (self.batch, self.m, 3):
            raise IcpAmdError(_lib.EINVAL, "expected P %r and Q %r" % ((self.batch, self.n, 3), (self.batch, self.m, 3)))
        check(lib().icp_amd_batch_upload(self._h, _p(P), _p(Q)))

    def run(self, iters, search="auto", timed=False):
        ms = C.c_double()
        check(lib().icp_amd_batch_run(self._h, C.c_uint(iters), C.c_int(SEARCH[search]), C.byref(ms) if timed else None))
        return ms.value if timed else None

    def download(self):
        b, n = self.batch, self.n
        d = dict(P_out=np.empty((b, n, 3)), R=np.empty((b, 3, 3)), t=np.empty((b, 3)), cov=np.empty((b, 3, 3)),
                 idx=np.empty((b, n), dtype=np.int32), rms=np.empty(b))
        check(lib().icp_amd_batch_download(self._h, _p(d["P_out"]), _p(d["R"]), _p(d["t"]), _p(d["cov"]), _p(d["idx"]), _p(d["rms"])))
        return d


def icp_gpu_batch(P, Q, iterations):
    """`batch` independent registrations: P [batch, n, 3], Q [batch, m, 3] -> dict(P_out, R, t, cov, idx, rms)."""
    P = np.ascontiguousarray(P, dtype=np.float64)
    Q = np.ascontiguousarray(Q, dtype=np.float64)
    if P.ndim != 3 or Q.ndim != 3 or P.shape[0] != Q.shape[0] or P.shape[2] != 3 or Q.shape[2] != 3:
        raise IcpAmdError(_lib.EINVAL, "expected P [batch, n, 3] and Q [batch, m, 3]")
    b = Batch(P.shape[0], P.shape[1], Q.shape[1])
    try:
        b.upload(P, Q)
        b.run(iterations)
        return b.download()
    finally:
        b.close()


def run_batch(P, Q, iterations):
    """icp_amd_run_batch: like icp_gpu_batch, any cloud size (large clouds go through ordinary engines, several at a time)."""
    P = np.ascontiguousarray(P, dtype=np.float64)
    Q = np.ascontiguousarray(Q, dtype=np.float64)
    if P.ndim != 3 or Q.ndim != 3 or P.shape[0] != Q.shape[0] or P.shape[2] != 3 or Q.shape[2] != 3:
        raise IcpAmdError(_lib.EINVAL, "expected P [batch, n, 3] and Q [batch, m, 3]")
    b, n, m = P.shape[0], P.shape[1], Q.shape[1]
    d = dict(P_out=np.empty_like(P), R=np.zeros((b, 3, 3)), t=np.zeros((b, 3)), idx=np.zeros((b, n), dtype=np.int32), rms=np.zeros(b))
    check(lib().icp_amd_run_batch(_p(P), _p(Q), C.c_size_t(b), C.c_size_t(n), C.c_size_t(m), C.c_uint(iterations), _p(d["P_out"]),
                                  _p(d["R"]), _p(d["t"]), _p(d["idx"]), _p(d["rms"])))
    return d


# ---- stage level (parity tests) --------------------------------------------------------------------
def correspondences(P, Pc, Qz, search="exact", q_splits=1):
    P, Qz = _f64(P), _f64(Qz)
    Pc = np.ascontiguousarray(Pc, dtype=np.float64)
    idx = np.empty(P.shape[0], dtype=np.int32)
    check(lib().icp_amd_correspondences(_p(P), C.c_size_t(P.shape[0]), _p(Pc), _p(Qz), C.c_size_t(Qz.shape[0]),
                                        C.c_int(SEARCH[search]), C.c_int(q_splits), _p(idx)))
    return idx


def correspondences_seeded(P, Pc, Qz, seed, search="f32", q_splits=1):
    """icp_amd_correspondences_seeded: the same search started from candidate correspondences (the seeded kernels)."""
    P, Qz = _f64(P), _f64(Qz)
    Pc = np.ascontiguousarray(Pc, dtype=np.float64)
    seed = np.ascontiguousarray(seed, dtype=np.int32)
    if seed.shape != (P.shape[0],):
        raise IcpAmdError(_lib.EINVAL, "expected one seed per source row")
    idx = np.empty(P.shape[0], dtype=np.int32)
    check(lib().icp_amd_correspondences_seeded(_p(P), C.c_size_t(P.shape[0]), _p(Pc), _p(Qz), C.c_size_t(Qz.shape[0]),
                                               C.c_int(SEARCH[search]), C.c_int(q_splits), _p(seed), _p(idx)))
    return idx


def cross_cov(P, Pc, Qz, idx):
    P, Qz = _f64(P), _f64(Qz)
    Pc = np.ascontiguousarray(Pc, dtype=np.float64)
    idx = np.ascontiguousarray(idx, dtype=np.int32)
    cov = np.empty((3, 3))
    check(lib().icp_amd_cross_cov(_p(P), C.c_size_t(P.shape[0]), _p(Pc), _p(Qz), C.c_size_t(Qz.shape[0]), _p(idx), _p(cov)))
    return cov


def centroid(A):
    A = _f64(A)
    out = np.empty(3)
    check(lib().icp_amd_centroid(_p(A), C.c_size_t(A.shape[0]), _p(out)))
    return out


def centroid_static(a_dev_ptr, n, out_dev_ptr, device=0, stream=None):
    """icp_amd_centroid_static: the centroid of n x 3 device-resident rows into 3 device doubles, enqueued on `stream`,
    without any allocation (static device scratch).  Pointers are raw device addresses (e.g. torch.Tensor.data_ptr())."""
    check(lib().icp_amd_centroid_static(C.c_void_p(a_dev_ptr), C.c_size_t(n), C.c_void_p(out_dev_ptr), C.c_int(device),
                                        C.c_void_p(stream)))


def centroid_static_max_rows():
    f = lib().icp_amd_centroid_static_max_rows
    f.restype = C.c_size_t
    return int(f())


def solve(cov, Pc, Qc):
    cov = np.ascontiguousarray(cov, dtype=np.float64).reshape(9)
    Pc = np.ascontiguousarray(Pc, dtype=np.float64)
    Qc = np.ascontiguousarray(Qc, dtype=np.float64)
    R, t, S = np.empty((3, 3)), np.empty(3), np.empty(3)
    check(lib().icp_amd_solve(_p(cov), _p(Pc), _p(Qc), _p(R), _p(t), _p(S)))
    return R, t, S


def apply(P, R, t):
    P = _f64(P)
    R = np.ascontiguousarray(R, dtype=np.float64).reshape(9)
    t = np.ascontiguousarray(t, dtype=np.float64)
    out = np.empty_like(P)
    check(lib().icp_amd_apply(_p(P), C.c_size_t(P.shape[0]), _p(R), _p(t), _p(out)))
    return out


# ---- host I/O kept from the reference ---------------------------------------------------------------
def read_csv(path):
    """readCSV (reference src/libCSV/csv.cpp:94-158): returns (array[nblines, nbcols], header)."""
    out = C.POINTER(C.c_double)()
    nl, nc = C.c_size_t(), C.c_size_t()
    hdr = C.create_string_buffer(4096)
    check(lib().icp_amd_read_csv(str(path).encode(), C.byref(out), C.byref(nl), C.byref(nc), hdr, C.c_size_t(4096)))
    try:
        a = np.ctypeslib.as_array(out, shape=(nl.value * nc.value,)).copy() if nl.value * nc.value else np.empty(0)
    finally:
        lib().icp_amd_free(out)
    return a.reshape(nl.value, nc.value), hdr.value.decode(errors="replace")


def _read(fn, path, *extra):
    out = C.POINTER(C.c_double)()
    nl, nc = C.c_size_t(), C.c_size_t()
    hdr = C.create_string_buffer(4096)
    check(fn(str(path).encode(), *extra, C.byref(out), C.byref(nl), C.byref(nc), hdr, C.c_size_t(4096)))
    try:
        a = np.ctypeslib.as_array(out, shape=(nl.value * nc.value,)).copy() if nl.value * nc.value else np.empty(0)
    finally:
        lib().icp_amd_free(out)
    return a.reshape(nl.value, nc.value), hdr.value.decode(errors="replace")


def read_csv_mt(path, threads=0):
    """icp_amd_read_csv with the lines spread over host threads: same values, shapes and errors."""
    return _read(lib().icp_amd_read_csv_mt, path, C.c_int(threads))


def read_cloud(path):
    """*.npy / *.f64 (raw fp64 x 3) / CSV by file name."""
    return _read(lib().icp_amd_read_cloud, path)


def write_cloud_binary(path, A):
    A = np.ascontiguousarray(A, dtype=np.float64)
    check(lib().icp_amd_write_cloud_binary(str(path).encode(), _p(A), C.c_size_t(A.shape[0]), C.c_size_t(A.shape[1])))


def write_cloud(path, A):
    A = np.ascontiguousarray(A, dtype=np.float64)
    check(lib().icp_amd_write_cloud(str(path).encode(), _p(A), C.c_size_t(A.shape[0]), C.c_size_t(A.shape[1])))


def euclidian_distance(A, B):
    A = np.ascontiguousarray(A, dtype=np.float64)
    B = np.ascontiguousarray(B, dtype=np.float64)
    if A.shape != B.shape:
        raise IcpAmdError(_lib.EINVAL, "element_wise_reduce requires arrays to be broadcastable")
    return float(lib().icp_amd_euclidian_distance(_p(A), _p(B), C.c_size_t(A.shape[0]), C.c_size_t(A.shape[1])))
