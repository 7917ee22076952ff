"""
ctypes binding of libgalois_amd.so -- the C-ABI declared in include/galois_amd.h.

The library is the product's only compute path.  If it is missing this module raises ImportError (no fallback):
build it with `python galois_amd/build.py` (hipcc, gfx950).
"""
from __future__ import annotations

import ctypes
import os

# PyTorch-ROCm must be loaded BEFORE the extension: both sides then share ONE HIP runtime instance (the one torch
# ships), so that torch's device pointers and stream handles are valid inside libgalois_amd.so.  Loading the extension
# first would pull in /opt/rocm's libamdhip64 as a second, separate runtime.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GALOIS_AMD_LIB") or os.path.join(_HERE, "libgalois_amd.so")  # (override: tuning builds only)

OK, ERR_INVALID, ERR_UNSUPPORTED, ERR_HIP, ERR_NOMEM = 0, 1, 2, 3, 4
U8, U16, U32, U64 = 0, 1, 2, 3
OP_ADD, OP_SUB, OP_MUL, OP_DIV, OP_NEG, OP_RECIP, OP_POW = range(7)
MODE_AUTO, MODE_LOOKUP, MODE_CALCULATE = 0, 1, 2
DEVERR_ZERO_DIVISION = 1
DEVERR_NO_LU = 2
DEVERR_LOG_ZERO = 4
DEVERR_LOG_BASE = 8

c_void_p, c_int, c_i64, c_u64, c_u32 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_uint64, ctypes.c_uint32
_u64p = ctypes.POINTER(ctypes.c_uint64)
_i64p = ctypes.POINTER(ctypes.c_int64)
_f32p = ctypes.POINTER(ctypes.c_float)

# name -> (restype, argtypes); must list every symbol declared in include/galois_amd.h
SIGNATURES = {
    "gfa_abi_version": (c_int, []),
    "gfa_last_error": (ctypes.c_char_p, []),
    "gfa_device_count": (c_int, []),
    "gfa_trim_scratch": (c_int, [c_u64]),
    "gfa_field_create": (c_int, [c_u64, c_u32, _u64p, c_u64, ctypes.POINTER(c_void_p)]),
    "gfa_field_destroy": (None, [c_void_p]),
    "gfa_field_set_mode": (c_int, [c_void_p, c_int]),
    "gfa_field_get_mode": (c_int, [c_void_p]),
    "gfa_field_order": (c_u64, [c_void_p]),
    "gfa_field_tables": (c_int, [c_void_p, _i64p, _i64p, _i64p, _i64p]),
    "gfa_scalar": (c_int, [c_void_p, c_int, c_u64, c_u64, _u64p]),
    "gfa_binary": (c_int, [c_void_p, c_int, c_void_p, c_i64, c_void_p, c_i64, c_void_p, c_i64, c_int, c_void_p, c_void_p]),
    "gfa_unary": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_i64, c_int, c_void_p, c_void_p]),
    "gfa_power": (c_int, [c_void_p, c_void_p, c_i64, c_void_p, c_i64, c_void_p, c_i64, c_int, c_void_p, c_void_p]),
    "gfa_scalar_multiply": (c_int, [c_void_p, c_void_p, c_i64, c_void_p, c_i64, c_void_p, c_i64, c_int, c_void_p]),
    "gfa_reduce": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_i64, c_i64, c_int, c_void_p, c_void_p]),
    "gfa_reduceat": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_i64, c_void_p, c_int, c_void_p, c_void_p]),
    "gfa_accumulate": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_i64, c_i64, c_int, c_void_p, c_void_p]),
    "gfa_convolve": (c_int, [c_void_p, c_void_p, c_i64, c_void_p, c_i64, c_void_p, c_int, c_void_p]),
    "gfa_ntt": (c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_i64, c_u64, c_int, c_int, c_void_p]),
    "gfa_wfield_create": (c_int, [c_int, c_u32, _u64p, ctypes.POINTER(c_void_p)]),
    "gfa_wfield_destroy": (None, [c_void_p]),
    "gfa_wide_binary": (c_int, [c_void_p, c_int, c_void_p, c_i64, c_void_p, c_i64, c_void_p, c_i64, c_void_p, c_void_p]),
    "gfa_wide_unary": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_i64, c_void_p, c_void_p]),
    "gfa_wide_power": (c_int, [c_void_p, c_void_p, c_i64, c_void_p, c_i64, c_void_p, c_void_p, c_i64, c_void_p, c_void_p]),
    "gfa_wide_reduce": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_i64, c_i64, c_int, c_void_p, c_void_p]),
    "gfa_wide_convolve": (c_int, [c_void_p, c_void_p, c_i64, c_void_p, c_i64, c_void_p, c_void_p]),
    "gfa_wide_matmul": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_void_p]),
    "gfa_bfield_create": (c_int, [c_int, c_u32, c_u32, _u64p, ctypes.POINTER(c_void_p)]),
    "gfa_bfield_destroy": (None, [c_void_p]),
    "gfa_big_binary": (c_int, [c_void_p, c_int, c_void_p, c_i64, c_void_p, c_i64, c_void_p, c_i64, c_void_p, c_void_p]),
    "gfa_big_unary": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_i64, c_void_p, c_void_p]),
    "gfa_big_power": (c_int, [c_void_p, c_void_p, c_i64, c_void_p, c_i64, c_void_p, c_void_p, c_i64, c_void_p, c_void_p]),
    "gfa_wide_row_reduce": (c_int, [c_void_p, c_void_p, c_i64, c_i64, c_i64, c_i64, c_void_p, c_void_p]),
    "gfa_wide_plu_decompose": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_i64, c_i64, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "gfa_wide_poly_evaluate": (c_int, [c_void_p, c_void_p, c_i64, c_void_p, c_void_p, c_i64, c_void_p]),
    "gfa_ntt_chunked": (c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_i64, c_u64, c_int, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_int, c_void_p]),
    "gfa_ntt_dist": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_i64, c_i64, c_u64, c_int, c_void_p]),
    "gfa_intt_dist": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_i64, c_i64, c_u64, c_int, c_int, c_void_p]),
    "gfa_ntt_columns": (c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_i64, c_i64, c_i64, c_u64, c_int, c_void_p]),
    "gfa_ntt_columns_pitched": (c_int, [c_void_p, c_void_p, c_i64, c_void_p, c_i64, c_i64, c_i64, c_i64, c_i64, c_u64, c_int, c_void_p]),
    "gfa_ntt_columns_inv": (c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_i64, c_i64, c_i64, c_u64, c_int, c_int, c_void_p]),
    "gfa_berlekamp_massey": (c_int, [c_void_p, c_void_p, c_i64, c_i64, c_void_p, c_void_p, c_int, c_void_p]),
    "gfa_vector": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_i64, c_void_p]),
    "gfa_poly_evaluate": (c_int, [c_void_p, c_void_p, c_i64, c_void_p, c_void_p, c_i64, c_int, c_void_p]),
    "gfa_log_prepare": (c_int, [c_void_p, _u64p, ctypes.POINTER(c_u32), c_int]),
    "gfa_log": (c_int, [c_void_p, c_void_p, c_i64, c_void_p, c_i64, c_void_p, c_i64, c_int, c_void_p, c_void_p]),
    "gfa_matmul": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_int, c_void_p]),
    "gfa_row_reduce": (c_int, [c_void_p, c_void_p, c_i64, c_i64, c_i64, c_i64, c_void_p, c_int, c_void_p]),
    "gfa_plu_decompose": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_i64, c_i64, c_int, c_void_p, c_void_p,
                                  c_int, c_void_p, c_void_p]),
    "gfa_time_matmul": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_i64, c_i64, c_i64, c_int, c_void_p, c_int,
                                ctypes.POINTER(ctypes.c_float)]),
    "gfa_rs_create": (c_int, [c_void_p, c_i64, c_i64, c_i64, c_u64, c_int, ctypes.POINTER(c_void_p)]),
    "gfa_bch_create": (c_int, [c_void_p, c_u64, c_i64, c_i64, c_i64, c_i64, c_u64, _u64p, c_int, ctypes.POINTER(c_void_p)]),
    "gfa_rs_destroy": (None, [c_void_p]),
    "gfa_rs_describe": (c_int, [c_void_p, _u64p, _u64p, _u64p]),
    "gfa_rs_encode": (c_int, [c_void_p, c_void_p, c_i64, c_void_p, c_i64, c_int, c_int, c_void_p]),
    "gfa_rs_extract_message": (c_int, [c_void_p, c_void_p, c_i64, c_void_p, c_i64, c_int, c_void_p]),
    "gfa_rs_detect": (c_int, [c_void_p, c_void_p, c_i64, c_void_p, c_i64, c_int, c_void_p]),
    "gfa_rs_decode": (c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_void_p, c_void_p, c_i64, c_int, c_void_p]),
    "gfa_time_binary": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_i64, c_int, c_void_p, c_int, _f32p]),
    "gfa_time_unary": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_i64, c_int, c_void_p, c_int, _f32p]),
    "gfa_time_ntt": (c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_i64, c_u64, c_int, c_void_p, c_int, _f32p]),
    "gfa_time_rs_encode": (c_int, [c_void_p, c_void_p, c_i64, c_void_p, c_i64, c_int, c_void_p, c_int, _f32p]),
    "gfa_time_rs_decode": (c_int, [c_void_p, c_void_p, c_i64, c_void_p, c_void_p, c_i64, c_int, c_void_p, c_int, _f32p]),
    "gfa_debug_fermat_stamps": (None, [c_void_p]),
    "gfa_debug_m32_tune": (None, [c_int, c_int]),
    "gfa_debug_rs_bm_selftest": (c_int, [c_void_p, c_i64, c_u64, _i64p, c_void_p]),
}

_lib = None


def lib() -> ctypes.CDLL:
    """Loads the library once.  Raises ImportError if it has not been built -- there is no CPU path."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: the HIP extension is the only compute path of galois_amd. "
                "Build it with `python galois_amd/build.py` (hipcc --offload-arch=gfx950)."
            )
        L = ctypes.CDLL(LIB_PATH)
        for name, (restype, argtypes) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the symbol is not exported
            fn.restype = restype
            fn.argtypes = argtypes
        if L.gfa_abi_version() != 1:
            raise ImportError("libgalois_amd.so ABI version mismatch")
        _lib = L
    return _lib


def last_error() -> str:
    msg = lib().gfa_last_error()
    return msg.decode() if msg else ""


class GfaError(RuntimeError):
    pass


def check(rc: int, what: str = ""):
    """Maps a gfa_status to the exception type the reference raises for the same condition."""
    if rc == OK:
        return
    msg = f"{what}: {last_error()}" if what else last_error()
    if rc == ERR_INVALID:
        raise ValueError(msg)
    if rc == ERR_UNSUPPORTED:
        raise NotImplementedError(msg)
    if rc == ERR_NOMEM:
        raise MemoryError(msg)
    raise GfaError(msg)
