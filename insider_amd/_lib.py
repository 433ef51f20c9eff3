"""ctypes binding of libinsider_hip.so (the C ABI declared in include/insider_hip.h).

There is no CPU fallback: every compute entry point raises InsiderError when the
shared library is missing or no HIP device is visible.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("INSIDER_HIP_LIB") or os.path.join(_HERE, "libinsider_hip.so")   # INSIDER_HIP_LIB: another build of the same C ABI

OK, ERR_ARG, ERR_SOLVE, ERR_ALLOC, ERR_HIP, ERR_NO_DEVICE, ERR_UNSUPPORTED, ERR_COMM = range(8)
TRAJ_STRIDE = 10
MAX_K = 63

ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)

# every symbol include/insider_hip.h declares (tests check the library exports all of them)
SYMBOLS = (
    "insider_hip_version", "insider_hip_last_error", "insider_hip_device_count", "insider_hip_create",
    "insider_hip_create_ex",
    "insider_hip_destroy", "insider_hip_set_shard", "insider_hip_set_option", "insider_hip_optimize",
    "insider_hip_optimize_oneshot", "insider_hip_optimize_row", "insider_hip_optimize_col", "insider_hip_strong_cd", "insider_hip_masked_gram_cols",
    "insider_hip_masked_gram_rows", "insider_hip_get_profile", "insider_hip_get_sweeps", "insider_hip_last_cd_ms",
    "insider_hip_optimize_oneshot_ex", "insider_hip_strong_cd_xy", "insider_hip_solve_sympd", "insider_hip_get_info",
    "insider_hip_comm_unique_id", "insider_hip_comm_init", "insider_hip_get_array", "insider_hip_clone",
    "insider_hip_optimize_continuous_v2",
)
COMM_ID_BYTES = 128


class InsiderError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"insider_hip status {status}: {message}")
        self.status = status


_lib = None


def load():
    """Load libinsider_hip.so (built in-tree by __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.environ.get("INSIDER_HIP_LIB"):
        # the in-tree library must be THE build of the sources on disk: compared by content hash (the library carries the
        # hash of what it was compiled from), never by mtime; a stale or missing binary is rebuilt, not run
        from . import _build
        if _build.needs_build(LIB_PATH):
            try:
                _build.build_library()
            except Exception as e:
                raise InsiderError(ERR_NO_DEVICE, f"{LIB_PATH} is missing or was built from other sources (library "
                                                  f"{_build.library_sha(LIB_PATH)}, sources {_build.source_sha()}) and the "
                                                  f"rebuild failed: {e!r} (there is no CPU fallback)")
    if not os.path.exists(LIB_PATH):
        raise InsiderError(ERR_NO_DEVICE, f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; "
                                          f"g.build()'` (there is no CPU fallback)")
    # Concurrent fits on one GPU (insider_hip_clone, tune(concurrent=k)) need their streams on DIFFERENT hardware queues: the
    # HIP runtime multiplexes all streams of a process onto GPU_MAX_HW_QUEUES queues (default 4), and two fits whose main
    # streams share a queue run one after the other.  Read by the runtime when it initialises: set before the first HIP call.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
    lib = C.CDLL(LIB_PATH)
    dp, i32p, u8p = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_uint8)
    lib.insider_hip_version.restype = C.c_char_p
    lib.insider_hip_last_error.restype = C.c_char_p
    lib.insider_hip_device_count.restype = C.c_int
    lib.insider_hip_create.argtypes = [dp, C.c_int64, C.c_int64, i32p, C.c_int, i32p, u8p, u8p, C.c_int,
                                       C.POINTER(C.c_void_p)]
    lib.insider_hip_create_ex.argtypes = [dp, C.c_int64, C.c_int64, i32p, C.c_int, i32p, dp, C.c_int, u8p, u8p, C.c_int,
                                          C.POINTER(C.c_void_p)]
    lib.insider_hip_clone.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
    lib.insider_hip_destroy.argtypes = [C.c_void_p]
    lib.insider_hip_destroy.restype = None
    lib.insider_hip_set_shard.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int, ALLREDUCE_FN, C.c_void_p]
    lib.insider_hip_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_double]
    lib.insider_hip_optimize.argtypes = [C.c_void_p, C.POINTER(dp), dp, C.c_int, C.c_int, C.c_double, C.c_double,
                                         C.c_double, C.c_int, C.c_double, C.c_double, C.c_uint32, C.c_uint64, dp, dp,
                                         dp, dp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.insider_hip_optimize_oneshot.argtypes = [dp, C.c_int64, C.c_int64, C.POINTER(dp), dp, i32p, C.c_int, i32p,
                                                 u8p, u8p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double,
                                                 C.c_int, C.c_double, C.c_double, C.c_uint32, C.c_uint64, dp, dp, dp]
    lib.insider_hip_optimize_oneshot_ex.argtypes = [dp, C.c_int64, C.c_int64, C.POINTER(dp), dp, i32p, C.c_int, i32p, dp,
                                                    C.c_int, u8p, u8p, C.c_int, C.c_int, C.c_double, C.c_double,
                                                    C.c_double, C.c_int, C.c_double, C.c_double, C.c_uint32, C.c_uint64,
                                                    C.c_int, dp, dp, dp]
    lib.insider_hip_strong_cd_xy.argtypes = [dp, dp, C.c_int64, C.c_int, dp, C.c_double, C.c_double, dp, dp, C.c_double,
                                             C.c_uint64, C.c_uint32, C.c_int, C.c_int, C.c_int, dp, i32p]
    lib.insider_hip_solve_sympd.argtypes = [dp, dp, C.c_int, C.c_int64, C.c_int, dp, i32p]
    lib.insider_hip_optimize_continuous_v2.argtypes = [dp, C.c_int64, C.c_int64, u8p, dp, dp, C.c_int, dp, dp, C.c_double,
                                                       C.c_int, C.c_int]
    lib.insider_hip_get_info.argtypes = [C.c_void_p, C.c_char_p, dp]
    lib.insider_hip_get_array.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64]
    lib.insider_hip_comm_unique_id.argtypes = [C.c_void_p, C.c_int]
    lib.insider_hip_comm_init.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    lib.insider_hip_optimize_row.argtypes = [C.c_void_p, C.POINTER(dp), dp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int]
    lib.insider_hip_optimize_col.argtypes = [C.c_void_p, C.POINTER(dp), dp, C.c_int, C.c_int, C.c_double, C.c_double,
                                             C.c_int, C.c_double, C.c_uint64, C.c_uint32]
    lib.insider_hip_strong_cd.argtypes = [dp, dp, dp, C.c_int, C.c_int64, C.c_double, C.c_double, C.c_double,
                                          C.c_uint64, C.c_uint32, C.c_int, C.c_int, C.c_int, dp, i32p]
    lib.insider_hip_masked_gram_cols.argtypes = [C.c_void_p, dp, C.c_int, dp, dp]
    lib.insider_hip_masked_gram_rows.argtypes = [C.c_void_p, dp, C.c_int, dp, dp]
    lib.insider_hip_get_profile.argtypes = [C.c_void_p, dp]
    lib.insider_hip_get_sweeps.argtypes = [C.c_void_p, i32p]
    lib.insider_hip_last_cd_ms.restype = C.c_double
    _lib = lib
    return lib


def library_source_sha():
    """The source hash the LOADED library reports (insider_hip_version(): 'src:<sha16>')."""
    v = load().insider_hip_version().decode()
    return v.rsplit("src:", 1)[1] if "src:" in v else None


def check(status):
    if status != OK:
        raise InsiderError(status, load().insider_hip_last_error().decode(errors="replace"))


def device_count():
    return int(load().insider_hip_device_count())


def f64(a):
    return np.asfortranarray(a, dtype=np.float64)


def ptr(a, t=C.c_double):
    return a.ctypes.data_as(C.POINTER(t))
