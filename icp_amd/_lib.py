"""ctypes binding of libicp_amd.so (the C ABI in include/icp_amd.h).  No compute happens in Python and
there is no fallback: if the HIP library is missing, importing the binding fails loudly."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ICP_AMD_LIB") or os.path.join(_HERE, "libicp_amd.so")   # ICP_AMD_LIB: another build of the same ABI (A/B runs)

OK, EINVAL, ENODEV, EHIP, ENOMEM, EIO, ESTATE = 0, -1, -2, -3, -4, -5, -6
METHODS = {"-shared-loop": 0, "-shared": 1, "-loop": 2, "-batch": 3}
SEARCH = {"auto": 0, "exact": 1, "fma": 2, "f32": 3, "grid": 4}


class Trace(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("Pc", "idx", "cov", "S", "R", "t", "rms")]


class Options(C.Structure):
    _fields_ = [("device", C.c_int), ("stream", C.c_void_p), ("search", C.c_int), ("q_splits", C.c_int),
                ("kernel_timing", C.c_int), ("rank", C.c_int), ("world", C.c_int), ("n_global", C.c_size_t),
                ("cov_gather", C.c_void_p), ("cen_gather", C.c_void_p), ("strict_order", C.c_int), ("sort_source", C.c_int), ("reserved", C.c_int * 6)]


class IcpAmdError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("icp_amd error %d: %s" % (code, msg))
        self.code = code


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("%s not built: run `make` (hipcc --offload-arch=gfx950); there is no CPU fallback" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        L.icp_amd_last_error.restype = C.c_char_p
        L.icp_amd_euclidian_distance.restype = C.c_double
        L.icp_amd_rows_per_rank.restype = C.c_size_t
        L.icp_amd_chunks_per_rank.restype = C.c_size_t
        L.icp_amd_rows_per_rank.argtypes = [C.c_size_t, C.c_int]
        L.icp_amd_chunks_per_rank.argtypes = [C.c_size_t, C.c_int]
        _lib = L
    return _lib


def check(rc):
    if rc != OK:
        raise IcpAmdError(rc, lib().icp_amd_last_error().decode(errors="replace"))
