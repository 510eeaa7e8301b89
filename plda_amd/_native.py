"""plda_amd/_native.py -- ctypes binding of libplda_hip.so (include/plda_hip.h).

There is no fallback: if the shared library is missing, or lacks a symbol, or no
gfx950 device is usable, an exception is raised.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_HERE, "lib", "libplda_hip.so")
SO_DIAG_PATH = os.path.join(_HERE, "lib", "libplda_hip_diag.so")   # -DPLDA_DIAG=1: + the measurement arms (build.py --diag)

PLDA_OK = 0
PLDA_E_INVAL = -1
PLDA_E_ONE_SPEAKER = -2
PLDA_E_NUMERIC = -3
PLDA_E_NOT_FITTED = -4
PLDA_E_HIP = -5
PLDA_E_LABELS = -6
PLDA_E_CAPACITY = -7

_vp = C.c_void_p
_i32, _i64, _f64 = C.c_int32, C.c_int64, C.c_double

# name -> (restype, argtypes); pointers are passed as void* (host ndarray.ctypes.data
# or a device address), exactly the plain-pointer ABI of include/plda_hip.h
SIGNATURES = {
    "plda_abi_version": (C.c_int, []),
    "plda_build_flags": (C.c_int, []),
    "plda_create": (C.c_int, [C.c_int, C.POINTER(_vp)]),
    "plda_destroy": (C.c_int, [_vp]),
    "plda_last_error": (C.c_char_p, [_vp]),
    "plda_set_stream": (C.c_int, [_vp, _vp]),
    "plda_reset_stream": (C.c_int, [_vp]),
    "plda_synchronize": (C.c_int, [_vp]),
    "plda_fit": (C.c_int, [_vp, _vp, _i64, _i32, _vp, _i32]),
    "plda_fit_dev": (C.c_int, [_vp, _vp, _i64, _i32, _vp, _i64, _i32]),
    "plda_fit_stats_dev": (C.c_int, [_vp, _vp, _i64, _i32, _vp, _i64]),
    "plda_fit_get_stats_dev": (C.c_int, [_vp, _vp, _vp, _vp]),
    "plda_fit_em_dev": (C.c_int, [_vp, _vp, _vp, _i64, _vp, _i32, _i32]),
    "plda_lda_fit": (C.c_int, [_vp, _vp, _i64, _i32, _vp, _i32, _vp]),
    "plda_lda_fit_dev": (C.c_int, [_vp, _vp, _i64, _i32, _vp, _i64, _i32, _vp]),
    "plda_lda_dims": (C.c_int, [_vp, C.POINTER(_i64), C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i32)]),
    "plda_lda_get_model": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "plda_lda_set_model": (C.c_int, [_vp, _i32, _i64, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "plda_lda_predict": (C.c_int, [_vp, _vp, _i64, _i32, _i32, _vp]),
    "plda_lda_predict_dev": (C.c_int, [_vp, _vp, _i64, _i32, _vp]),
    "plda_lda_transform": (C.c_int, [_vp, _vp, _i64, _i32, _i32, _vp]),
    "plda_lda_transform_dev": (C.c_int, [_vp, _vp, _i64, _i32, _vp]),
    "plda_htk_frames": (C.c_int, [_vp, _vp, _i64, _vp, _vp, _i64, _i32, _i32, _vp]),
    "plda_htk_frames_dev": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i64, _i32, _i32, _vp]),
    "plda_fit_timings": (C.c_int, [_vp, _vp]),
    "plda_fit_plan": (C.c_int, [_vp, _vp]),
    "plda_fit_get_stats": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "plda_fit_num_classes": (C.c_int, [_vp, C.POINTER(_i64)]),
    "plda_get_dims": (C.c_int, [_vp, C.POINTER(_i32), C.POINTER(_i32)]),
    "plda_get_model": (C.c_int, [_vp, _vp, _vp, _vp, _vp]),
    "plda_set_model": (C.c_int, [_vp, _i32, _i32, _vp, _vp, _vp]),
    "plda_truncate": (C.c_int, [_vp, _i32]),
    "plda_smooth": (C.c_int, [_vp, _f64]),
    "plda_transform_groups": (C.c_int, [_vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp, C.POINTER(_i64)]),
    "plda_transform_groups_dev": (C.c_int, [_vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp, C.POINTER(_i64)]),
    "plda_transform_rows": (C.c_int, [_vp, _vp, _i64, _i32, _vp, _i32, _vp]),
    "plda_transform_rows_dev": (C.c_int, [_vp, _vp, _i64, _i32, _vp, _i32, _vp]),
    "plda_score_pairs": (C.c_int, [_vp, _vp, _vp, _i64, _vp, _i64, _vp, _vp, _i64, _vp, _vp, _vp]),
    "plda_score_one": (C.c_int, [_vp, _vp, _i32, _vp, _i32, _f64, _f64, C.POINTER(_f64)]),
    "plda_score_matrix": (C.c_int, [_vp, _vp, _vp, _i32, _i64, _vp, _i64, _vp, _vp, _vp, _i64]),
    "plda_score_matrix_dev": (C.c_int, [_vp, _vp, _vp, _i32, _i64, _vp, _i64, _vp, _vp, _vp, _i64]),
    "plda_score_prepare_dev": (C.c_int, [_vp, _vp, _i64, _i32, _i32]),
    "plda_score_prepare_counts_dev": (C.c_int, [_vp, _vp, _i64, _vp, _i32]),
    "plda_score_unprepare": (C.c_int, [_vp]),
    "plda_profile_enable": (C.c_int, [_vp, _i32]),
    "plda_profile_read": (C.c_int, [_vp, C.POINTER(_f64), C.POINTER(_i64), C.POINTER(_f64), _i32]),
    "plda_profile_timeline": (C.c_int, [_vp, _vp, _i64]),
    "plda_sym_eig": (C.c_int, [_vp, _vp, _i32, _i32, _vp, _vp, _vp]),
    "plda_spd_inverse": (C.c_int, [_vp, _vp, _i32, _vp]),
    "plda_gemm_f64": (C.c_int, [_vp, _i64, _i64, _i64, C.c_double, _vp, _i32, _vp, _i32, _vp, C.c_double, _vp, _i32]),
    "plda_trace_enable": (C.c_int, [_vp, _i32]),
    "plda_trace_read": (C.c_int, [_vp, _vp, _i64, _i32]),
    "plda_score_last_shape": (C.c_int, [_vp, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i32)]),
    "plda_score_last_kernel": (C.c_int, [_vp, C.c_char_p, _i64]),
    "plda_dvector_pool": (C.c_int, [_vp, _vp, _i32, _i64, _i32, _vp, _i64, _i32, _i32, _vp]),
    "plda_dvector_pool_dev": (C.c_int, [_vp, _vp, _i32, _i64, _i32, _vp, _i64, _i32, _i32, _vp]),
    "plda_eer_matrix_dev": (C.c_int, [_vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp]),
    "plda_eer_lists": (C.c_int, [_vp, _vp, _i64, _vp, _i64, _vp]),
    "plda_det_matrix_dev": (C.c_int, [_vp, _vp, _i64, _i64, _i64, _vp, _vp, _i32, _vp, _vp, _vp]),
    "plda_det_lists": (C.c_int, [_vp, _vp, _i64, _vp, _i64, _i32, _vp, _vp, _vp]),
    "plda_score_eer_dev": (C.c_int, [_vp, _vp, _vp, _i32, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _vp]),
    "plda_eer_matrix_sharded_dev": (C.c_int, [_vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp]),
    "plda_comm_unique_id": (C.c_int, [_vp, _i64]),
    "plda_comm_init": (C.c_int, [_vp, _i32, _i32, _vp]),
    "plda_comm_init_custom": (C.c_int, [_vp, _i32, _i32, _vp]),
    "plda_comm_init_host": (C.c_int, [_vp, _i32, _i32, _vp]),
    "plda_comm_init_peer": (C.c_int, [_vp, _i32, _i32, _vp]),
    "plda_comm_destroy": (C.c_int, [_vp]),
    "plda_comm_describe": (C.c_int, [_vp, _vp, _i64]),
    "plda_shard_plan": (C.c_int, [_i64, _i32, _i32, _i64, _vp, _vp, _i64, C.POINTER(_i64), C.POINTER(_i64)]),
    "plda_score_matrix_sharded_local_dev": (C.c_int, [_vp, _vp, _vp, _i32, _i64, _vp, _i64, _vp, _vp, _vp, _i64, _i64, _vp, _i64]),
    "plda_comm_emulate": (C.c_int, [_vp, _i32, _i32]),
    "plda_comm_info": (C.c_int, [_vp, C.POINTER(_i32), C.POINTER(_i32)]),
    "plda_score_matrix_sharded_dev": (C.c_int, [_vp, _vp, _vp, _i32, _i64, _vp, _i64, _vp, _vp, _vp, _i64, _i64, _i32]),
    "plda_znorm_stats_sharded_dev": (C.c_int, [_vp, _vp, _i64, _i32, _i32, _vp, _i64, _vp, _vp]),
    "plda_fit_sharded_dev": (C.c_int, [_vp, _vp, _i64, _i32, _vp, _i64, _i32]),
    "plda_eer_matrix_comm_dev": (C.c_int, [_vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp]),
    "plda_znorm_stats": (C.c_int, [_vp, _vp, _i64, _i32, _i32, _vp, _i64, _vp, _vp]),
    "plda_znorm_stats_dev": (C.c_int, [_vp, _vp, _i64, _i32, _i32, _vp, _i64, _vp, _vp]),
}

# plda_collectives / plda_host_collectives (include/plda_hip.h): callback tables of the multi-GPU entry points
PLDA_DT_F64, PLDA_DT_U64, PLDA_DT_U32 = 0, 1, 2
PLDA_OP_SUM, PLDA_OP_MAX, PLDA_OP_MIN = 0, 1, 2
HOST_ALL_GATHER_V = C.CFUNCTYPE(C.c_int, _vp, _vp, C.POINTER(_i64), C.POINTER(_i64))
HOST_ALL_REDUCE = C.CFUNCTYPE(C.c_int, _vp, _vp, _i64, _i32, _i32)
DESTROY_FN = C.CFUNCTYPE(None, _vp)
DEV_ALL_GATHER = C.CFUNCTYPE(C.c_int, _vp, _vp, _vp, _i64, _vp)
DEV_ALL_GATHER_V = C.CFUNCTYPE(C.c_int, _vp, _vp, C.POINTER(_i64), C.POINTER(_i64), _vp)
DEV_ALL_REDUCE = C.CFUNCTYPE(C.c_int, _vp, _vp, _i64, _i32, _i32, _vp)


class HostCollectives(C.Structure):
    _fields_ = [("ctx", _vp), ("all_gather_v", HOST_ALL_GATHER_V), ("all_reduce", HOST_ALL_REDUCE), ("destroy", DESTROY_FN)]


class Collectives(C.Structure):
    _fields_ = [("ctx", _vp), ("all_gather", DEV_ALL_GATHER), ("all_gather_v", DEV_ALL_GATHER_V),
                ("all_reduce", DEV_ALL_REDUCE), ("destroy", DESTROY_FN)]


_libs = {}


def load(diag=None):
    """Load libplda_hip.so and bind every symbol of include/plda_hip.h.  diag=True (or PLDA_LIB_DIAG=1 when diag is None): the
    diagnostic build libplda_hip_diag.so, which additionally holds the measurement arms (profiling scripts only)."""
    if diag is None:
        diag = os.environ.get("PLDA_LIB_DIAG", "0") == "1"
    diag = bool(diag)
    if diag in _libs:
        return _libs[diag]
    path = SO_DIAG_PATH if diag else SO_PATH
    if not os.path.exists(path):
        raise ImportError(
            "plda_amd: %s not found -- build it with `python -m plda_amd.build%s` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback." % (path, " --diag" if diag else ""))
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    if bool(lib.plda_build_flags() & 1) != diag:
        raise ImportError("plda_amd: %s was built with%s -DPLDA_DIAG=1" % (path, "out" if diag else ""))
    _libs[diag] = lib
    return lib


class PldaError(RuntimeError):
    def __init__(self, code, message):
        RuntimeError.__init__(self, "libplda_hip error %d: %s" % (code, message))
        self.code = code
        self.message = message


def last_error(handle):
    msg = load().plda_last_error(handle)
    return msg.decode("utf-8", "replace") if msg else ""


def check(handle, rc):
    if rc != PLDA_OK:
        raise PldaError(rc, last_error(handle))
