"""ctypes binding of libdfx.so (C ABI declared in include/dfx.h).

There is no fallback: if the shared library is missing or a call fails, this raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# DFX_LIB lets tools/tune_kernels.py load a build variant; the product default is the in-tree library
LIB_PATH = os.environ.get("DFX_LIB") or os.path.join(_HERE, "libdfx.so")


class Geom(C.Structure):
    _fields_ = [("nT", C.c_int64), ("nL", C.c_int64), ("nZ", C.c_int64), ("X", C.c_int64), ("Y", C.c_int64),
                ("r", C.c_int32), ("pad_", C.c_int32)]


class Plan(C.Structure):
    _fields_ = [("fast", C.c_int32), ("zc", C.c_int32), ("n_zchunks", C.c_int32), ("steps", C.c_int32),
                ("n_chunks", C.c_int32), ("cpb", C.c_int32), ("n_blocks", C.c_int32), ("zs", C.c_int32),
                ("u_part_elems", C.c_int64),
                ("v_part_elems", C.c_int64), ("part_elems", C.c_int64)]


# slots of the device state `scal` (include/dfx.h)
S_RSOLD, S_RSNEW, S_PAP, S_ALPHA, S_BETA, S_TOL, S_UU, S_PP, S_XX, S_XP, S_VV, S_RR, S_XXNEW, S_NONFINITE, \
    S_ITER, S_SUM, S_SCALE, S_LOSS1, S_LOSS2, S_ADAM_DD, S_ADAM_XX, S_RA2, S_RB2, S_RAB, S_ALPHA_LAST = range(25)
NSCAL = 32
NPART = 4
EW_MAXBLOCKS = 2048
MAX_SLICES = 16
NOALLOC_MAX_PS = 65536
SWEEP_PROJECT, SWEEP_XP, SWEEP_P, SWEEP_PINIT = range(4)
LAST_NO, LAST_FOLD, LAST_NOFOLD = range(3)
POST_U, POST_V, POST_V_SUMMED, POST_FOLD_SWEEP, POST_FOLD_ADAM, POST_ALPHA = 1, 2, 4, 8, 16, 32
ADAM_NONNEG, ADAM_NOFOLD = 1, 2

_P = C.c_void_p
_G = C.POINTER(Geom)
_PL = C.POINTER(Plan)
_f, _d, _i32, _i64 = C.c_float, C.c_double, C.c_int32, C.c_int64

# name -> argtypes; must list every function declared in include/dfx.h (tests check this)
SIGNATURES = {
    "dfx_version": [],
    "dfx_last_error": [],
    "dfx_plan_make": [_G, _i32, _PL],
    "dfx_plan_make_ex": [_G, _i32, _i32, _i32, _PL],
    "dfx_sweep": [_i32, _G, _PL, _P, _P, _P, _P, _P, _P, _P, _P],
    "dfx_bcast": [_G, _PL, _P, _P, _P, _P, _P, _f, _f, _f, _P, _P, _P, _P],
    "dfx_adam_step": [_G, _PL, _P, _P, _P, _P, _P, _P, _d, _d, _d, _d, _i32, _i32, _P, _P, _P, _P, _P],
    "dfx_adam_step_factored": [_G, _PL, _P, _P, _P, _P, _P, _P, _P, _d, _d, _d, _d, _i32, _i32, _P, _P, _P, _P, _P],
    "dfx_adam_step_chunks": [_G, _PL, _P, _P, _P, _P, _P, _P, _P, _P, _d, _d, _d, _d, _i32, _i32, _P, _P, _P, _P,
                             _i32, _i32, _i32, _P],
    "dfx_lerp": [_P, _P, _i64, _d, _P],
    "dfx_adam_apply": [_P, _P, _P, _P, _i64, _d, _d, _d, _d, _i32, _i32, _i32, _P, _P, _P],
    "dfx_adam_apply_project": [_G, _PL, _P, _P, _P, _P, _d, _d, _d, _d, _i32, _i32, _P, _P, _P, _P, _P],
    "dfx_adam_prep": [_P, _P, _P, _P, _P, _i64, _P, _P, _P, _i64, _P, _P, _i64, _d, _d, _d, _P, _P, _P],
    "dfx_reduce_u": [_G, _PL, _P, _P, _P, _P, _P],
    "dfx_reduce_v": [_G, _PL, _P, _P, _i64, _i64, _P],
    "dfx_sweep_chunks": [_i32, _G, _PL, _P, _P, _P, _P, _P, _P, _P, _i32, _i32, _i32, _i32, _P],
    "dfx_sweep_factored": [_i32, _G, _PL, _P, _P, _P, _P, _P, _P, _P, _P, _i32, _i32, _i32, _i32, _P],
    "dfx_resid_update": [_G, _P, _P, _P, _P, _f, _f, _f, _P, _i32, _P, _P, _P, _P],
    "dfx_resid_update_noalloc": [_G, _P, _P, _P, _P, _f, _f, _f, _P, _i32, _P, _P],
    "dfx_resid_cg_step": [_G, _P, _P, _P, _P, _d, _d, _d, _i32, _P, _P, _P, _P],
    "dfx_post_sweep": [_G, _PL, _P, _P, _P, _P, _P, _P, _i32, _i32, _d, _d, _d, _P],
    "dfx_sqnorm": [_P, _i64, _P, _i32, _P, _P],
    "dfx_sqdiff": [_P, _P, _P, _i64, _i64, _d, _P, _i32, _P, _P],
    "dfx_cg_alpha": [_P, _d, _d, _d, _P],
    "dfx_cg_beta": [_P, _d, _i32, _P],
    "dfx_cg_start": [_P, _P],
    "dfx_axpy": [_P, _P, _i64, _f, _P, _P, _P, _P],
    "dfx_scale": [_P, _i64, _P, _i64, _f, _P, _P],
    "dfx_sum": [_P, _i64, _P, _P, _P],
    "dfx_recip": [_P, _P, _d, _P],
    "dfx_diff_scale": [_P, _P, _P, _P, _i64, _i64, _f, _P],
    "dfx_sum_axis": [_P, _P, _i64, _i64, _i64, _P],
    "dfx_expand_axis": [_P, _P, _i64, _i64, _i64, _f, _P],
    "dfx_block_sum": [_P, _P, _i64, _i64, _i64, _i32, _P],
    "dfx_block_expand": [_P, _P, _i64, _i64, _i64, _i32, _f, _P],
    "dfx_baseline_init": [_G, _P, _P, _P, _f, _P, _P, _P, _P],
    "dfx_resize_stack": [_P, _P, _P, _P, _i64, _i64, _i64, _i64, _i64, _i32, _P],
    "dfx_arena_create": [_i64, _i64, C.POINTER(_P)],
    "dfx_arena_grow": [_P, _i64],
    "dfx_arena_info": [_P, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i32)],
    "dfx_arena_map": [_P, C.POINTER(_i32), _i64, C.POINTER(_P)],
    "dfx_arena_unmap": [_P, _P],
    "dfx_arena_release_chunks": [_P, C.POINTER(_i32), _i64],
    "dfx_arena_destroy": [_P],
    "dfx_arena_last_error": [],
}

_lib = None


class DfxError(RuntimeError):
    pass


def lib() -> C.CDLL:
    """Load libdfx.so once; raise loudly if it is not built (no CPU / eager fallback exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DfxError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). datafusion_amd has no fallback path.")
        _lib = C.CDLL(LIB_PATH)
        for name, args in SIGNATURES.items():
            fn = getattr(_lib, name)
            fn.argtypes = args
            fn.restype = C.c_char_p if name in ("dfx_last_error", "dfx_arena_last_error") else C.c_int
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().dfx_arena_last_error() if what.startswith("dfx_arena") else lib().dfx_last_error()
        raise DfxError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")


def ptr(t) -> int | None:
    """Raw device pointer of a torch tensor (None -> NULL)."""
    if t is None:
        return None
    assert t.is_contiguous(), "dfx kernels take contiguous tensors"
    return t.data_ptr()
