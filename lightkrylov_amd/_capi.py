"""ctypes binding of the C ABI in include/lightkrylov_hip.h.

The shared library is built in-tree (``lightkrylov_amd/liblightkrylov_hip.so``) by
``__graft_entry__.build()`` / ``make -C lightkrylov_amd/csrc``.  There is no fallback of any
kind: if the library is missing, or no HIP device is present, every entry point raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liblightkrylov_hip.so")

LK_F64, LK_C128 = 0, 1
LK_DGS_NORMALIZE = 1
LK_OP_N, LK_OP_H = 0, 1
LK_ACCESS_READ, LK_ACCESS_OVERWRITE, LK_ACCESS_READWRITE = 0, 1, 2
LK_COMM_ID_BYTES = 128

ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)
HALO_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)
PROGRESS_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int)
ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_int, C.c_void_p)

_p = C.c_void_p
_pp = C.POINTER(C.c_void_p)
_dp = C.POINTER(C.c_double)
_i64 = C.c_int64
_int = C.c_int
_ip = C.POINTER(C.c_int)

# name -> (restype, argtypes).  Mirrors include/lightkrylov_hip.h declaration by declaration.
SIGNATURES = {
    "lk_version": (_int, []),
    "lk_last_error": (C.c_char_p, []),
    "lk_init": (_int, [_int, _p, _pp]),
    "lk_finalize": (_int, [_p]),
    "lk_sync": (_int, [_p]),
    "lk_set_allreduce": (_int, [_p, ALLREDUCE_FN, _p, _int, _int]),
    "lk_set_halo_exchange": (_int, [_p, HALO_FN, _p]),
    "lk_set_allgather": (_int, [_p, ALLGATHER_FN, _p]),
    "lk_context_info": (_int, [_p, _ip, _pp]),
    "lk_comm_info": (_int, [_p, _ip, _ip]),
    "lk_comm_available": (_int, []),
    "lk_comm_get_unique_id": (_int, [_p]),
    "lk_comm_init_rank": (_int, [_p, _int, _int, _p]),
    "lk_comm_destroy": (_int, [_p]),
    "lk_set_partition": (_int, [_p, _i64, _i64]),
    "lk_set_tuning": (_int, [_p, C.c_char_p, _int]),
    "lk_lazy_stats": (_int, [_p, C.POINTER(_i64)]),
    "lk_lazy_fusion_stats": (_int, [_p, C.POINTER(_i64)]),
    "lk_lazy_speculation_stats": (_int, [_p, C.POINTER(_i64)]),
    "lk_resident_stats": (_int, [_p, C.POINTER(_i64)]),
    "lk_resident_phase_ticks": (_int, [_p, C.POINTER(_i64)]),
    "lk_profile_enable": (_int, [_p, _int]),
    "lk_profile_get": (_int, [_p, C.c_char_p, C.POINTER(_i64), _dp, _dp]),
    "lk_profile_reset": (_int, [_p]),
    "lk_basis_create": (_int, [_p, _int, _i64, _int, _pp]),
    "lk_basis_wrap": (_int, [_p, _int, _i64, _int, _i64, _p, _pp]),
    "lk_basis_destroy": (_int, [_p]),
    "lk_basis_info": (_int, [_p, _ip, C.POINTER(_i64), _ip, C.POINTER(_i64), _pp]),
    "lk_vec_device_ptr": (_int, [_p, _int, _int, _pp]),
    "lk_basis_upload": (_int, [_p, _int, _int, _p, _i64]),
    "lk_basis_download": (_int, [_p, _int, _int, _p, _i64]),
    "lk_pool_acquire": (_int, [_p, _int, _i64, C.c_uint64, _pp, _ip]),
    "lk_pool_owner": (_int, [_p, _p, _int, C.POINTER(C.c_uint64)]),
    "lk_pool_column_info": (_int, [_p, _p, _int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "lk_pool_release": (_int, [_p, _p, _int]),
    "lk_pool_release_all": (_int, [_p]),
    "lk_pool_stats": (_int, [_p, C.POINTER(_i64)]),
    "lk_vec_zero": (_int, [_p, _int]),
    "lk_vec_rand": (_int, [_p, _int, C.c_uint64, _i64, _int]),
    "lk_vec_scal": (_int, [_p, _int, _dp]),
    "lk_vec_axpby": (_int, [_dp, _p, _int, _dp, _p, _int]),
    "lk_vec_dot": (_int, [_p, _int, _p, _int, _dp]),
    "lk_vec_norm": (_int, [_p, _int, _dp]),
    "lk_vec_size": (_int, [_p, C.POINTER(_i64)]),
    "lk_vec_copy": (_int, [_p, _int, _p, _int]),
    "lk_innerprod": (_int, [_p, _int, _p, _int, _int, _dp]),
    "lk_lincomb": (_int, [_p, _int, _dp, _int, _p, _int]),
    "lk_gram": (_int, [_p, _int, _dp]),
    "lk_orthogonalize": (_int, [_p, _int, _p, _int, _dp, _ip]),
    "lk_dgs": (_int, [_p, _int, _p, _int, _dp, _dp, _int, _ip]),
    "lk_dgs_block": (_int, [_p, _int, _p, _int, _int, _dp, _ip]),
    "lk_linop_diag_create": (_int, [_p, _int, _i64, _p, _pp]),
    "lk_linop_diag_linspace_create": (_int, [_p, _i64, _i64, C.c_double, C.c_double, _pp]),
    "lk_linop_dense_create": (_int, [_p, _int, _i64, _p, _i64, _pp]),
    "lk_linop_dense_create_sharded": (_int, [_p, _int, _i64, C.POINTER(_i64), _p, _i64, _pp]),
    "lk_linop_dense_wrap_sharded": (_int, [_p, _int, _i64, C.POINTER(_i64), _p, _i64, _pp]),
    "lk_linop_csr_create": (_int, [_p, _int, _i64, _p, _p, _p, _pp]),
    "lk_linop_csr_create_sharded": (_int, [_p, _int, _i64, C.POINTER(_i64), _p, _p, _p, _pp]),
    "lk_linop_lap5_create": (_int, [_p, _i64, _pp]),
    "lk_linop_lap5_create_sharded": (_int, [_p, _i64, _i64, _i64, _pp]),
    "lk_linop_gl_create_sharded": (_int, [_p, _i64, _i64, _i64, C.c_double, C.c_double, _int, _dp, _dp, C.c_double, C.c_double, _pp]),
    "lk_linop_gl_create": (_int, [_p, _i64, C.c_double, C.c_double, _int, _dp, _dp, C.c_double, C.c_double, _pp]),
    "lk_linop_destroy": (_int, [_p]),
    "lk_linop_apply": (_int, [_p, _int, _p, _int, _p, _int]),
    "lk_arnoldi": (_int, [_p, _p, _dp, _i64, _int, _int, C.c_double, _int, _ip]),
    "lk_arnoldi_segments": (_int, [_p, _p, _dp, _i64, _int, _int, C.c_double, _int, _ip, _int, PROGRESS_FN, _p, _ip]),
    "lk_lanczos": (_int, [_p, _p, _dp, _i64, _int, _int, C.c_double, _ip]),
    "lk_bidiag": (_int, [_p, _p, _p, _dp, _i64, _int, _int, C.c_double, _ip]),
    "lk_qr": (_int, [_p, _int, _int, _dp, _i64, C.c_double, _ip]),
    "lk_arnoldi_block": (_int, [_p, _p, _dp, _i64, _int, _int, _int, C.c_double, _int, _ip]),
}


class LightKrylovHipError(RuntimeError):
    """Raised for any non-zero status of the C ABI (the reference would `stop_error`)."""


_lib = None


def load():
    """Load the HIP engine.  Raises (never falls back) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise LightKrylovHipError(
            f"HIP engine not built: {LIB_PATH} is missing. Run `python -c 'import __graft_entry__ as g; "
            "g.build()'` or `make -C lightkrylov_amd/csrc`. There is no CPU fallback.")
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64/libhsa-runtime64.  Import
    # torch FIRST so this library's DT_NEEDED libamdhip64.so resolves to the copy torch already
    # loaded (same SONAME); loading ours first puts two HSA runtimes in the process and the second
    # one finds no device.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int) -> None:
    if rc != 0:
        msg = load().lk_last_error()
        raise LightKrylovHipError(f"[{rc}] {msg.decode() if msg else 'unknown error'}")
