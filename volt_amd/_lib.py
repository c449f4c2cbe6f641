"""ctypes binding of libvolt_hip.so (include/volt_hip.h).  No fallback: if the library is missing
or a call fails, this raises -- the product path never routes around the HIP kernels."""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  -- load torch's libamdhip64 first so the library binds the same HIP runtime

from .build import LIB, build_lib, source_hash

_lib = None
ABI_VERSION = 2

_i32, _i64, _f32, _f64, _ptr, _sz = C.c_int, C.c_int64, C.c_float, C.c_double, C.c_void_p, C.c_size_t

_SIGS = {
    "volt_abi_version": (C.c_int, []),
    "volt_source_hash": (C.c_char_p, []),
    "volt_padded_n": (C.c_int, [_i32]),
    "volt_cumtrapz_f32": (C.c_int, [_ptr, _i64, _ptr, _i64, _ptr, _i32, _i32, _i32, _ptr]),
    "volt_cumtrapz_f64": (C.c_int, [_ptr, _i64, _ptr, _i64, _ptr, _i32, _i32, _i32, _ptr]),
    "volt_fill_f32": (C.c_int, [_ptr, _ptr, _i32, _i32, _i64, _i64, _ptr]),
    "volt_fill_f64": (C.c_int, [_ptr, _ptr, _i32, _i32, _i64, _i64, _ptr]),
    "volt_ewma_f32": (C.c_int, [_ptr, _i64, _ptr, _i32, _ptr, _i32, _i32, _ptr]),
    "volt_prepare_f32": (C.c_int, [_ptr, _i64, _i64, _ptr, _f32, _ptr, _i32, _i32, _ptr]),
    "volt_potrf_f32": (C.c_int, [_ptr, _ptr, _ptr, _i32, _i32, _ptr]),
    "volt_potrf_workspace_bytes": (C.c_size_t, [_i32, _i32]),
    "volt_potrf_ws_f32": (C.c_int, [_ptr, _ptr, _ptr, _i32, _i32, _ptr, C.c_size_t, _ptr]),
    "volt_prepare_f64": (C.c_int, [_ptr, _i64, _i64, _ptr, _f64, _ptr, _i32, _i32, _ptr]),
    "volt_potrf_f64": (C.c_int, [_ptr, _ptr, _ptr, _i32, _i32, _ptr]),
    "volt_trsv_lower_f64": (C.c_int, [_ptr, _ptr, _ptr, _ptr, _ptr, _i32, _i32, _ptr]),
    "volt_trsv_lower_t_f64": (C.c_int, [_ptr, _ptr, _ptr, _ptr, _ptr, _i32, _i32, _ptr]),
    "volt_trsv_lower_f32": (C.c_int, [_ptr, _ptr, _ptr, _ptr, _ptr, _i32, _i32, _ptr]),
    "volt_trsv_lower_t_f32": (C.c_int, [_ptr, _ptr, _ptr, _ptr, _ptr, _i32, _i32, _ptr]),
    "volt_trtri_f32": (C.c_int, [_ptr, _ptr, _ptr, _i32, _i32, _ptr]),
    "volt_rollout_scratch_bytes": (_sz, [_i32, _i32, _i32]),
    "volt_rollout_bordered_f32": (C.c_int, [_ptr] * 16 + [_i32] * 6 + [_f32] * 3 + [_ptr]),
    "volt_profile_factor_f32": (C.c_int, [_ptr, _i64, _i64, _ptr, _ptr, _ptr, _ptr, _ptr, _i32, _i32, _i32, _ptr, _ptr, _ptr,
                                          _ptr, _ptr]),
    "volt_tune_update_f32": (C.c_int, [_ptr, _ptr, _ptr, _i32, _i32, _i32, _i32, _i32, _ptr]),
    "volt_adam_step_f32": (C.c_int, [_ptr, _i32, C.c_longlong, _ptr, C.c_float, C.c_float, C.c_float, C.c_float, _ptr, _ptr]),
    "volt_sched_describe": (C.c_int, [_i32, _i32, _i32, _i32, _i32, _i32, C.c_float, _ptr, _i32, _ptr]),
    "volt_tune_diag_f32": (C.c_int, [_ptr, _ptr, _ptr, _i32, _i32, _i32, _ptr, _ptr]),
    "volt_mll_workspace_bytes": (_sz, [_i32, _i32, _i32]),
    "volt_mll_step_f32": (C.c_int, [_ptr, _i64, _i64, _ptr, _ptr, _f32, _ptr, _ptr, _ptr, _ptr, _i32, _i32, _i32, _ptr]),
    "volt_mll_grad_k_f32": (C.c_int, [_ptr, _ptr, _ptr, _ptr, _i32, _i32, _ptr]),
    "volt_rollout_shared_f32": (C.c_int, [_ptr] * 8 + [_i32] * 5 + [_f32, _ptr]),
    "volt_gemm_nt_f32": (C.c_int, [_ptr, _i64, _i64, _i32, _ptr, _i64, _i64, _i32, _ptr, _i64, _i64, _i32, _f32, _f32,
                                   _i32, _i32, _i32, _i32, _ptr]),
    "volt_gpcv_workspace_bytes": (_sz, [_i32, _i32, _i32]),
    "volt_gpcv_step_f32": (C.c_int, [_ptr, _i64, _i64, _f32] + [_ptr] * 6 + [_i32, _f32, _f32, _f32, _f32] + [_ptr] * 7
                           + [_i32, _i32, _ptr]),
}

EXPORTS = tuple(_SIGS)


class VoltHipError(RuntimeError):
    pass


def lib() -> C.CDLL:
    """Load (once) and type the library.  Raises if it has not been built."""
    global _lib
    if _lib is None:
        # A source-only checkout, or a git-ignored binary older than the checked-in sources: (re)build once
        # (hipcc, ~30 s).  build_lib compares the hash of the sources with the one recorded by the last build.
        try:
            build_lib(verbose=False)
        except Exception as e:
            raise VoltHipError(
                f"{LIB} is missing or stale and could not be built ({e}): run `python -m volt_amd.build` "
                "(hipcc, gfx950). volt_amd has no CPU fallback.") from e
        handle = C.CDLL(LIB)
        built = handle.volt_source_hash
        built.restype = C.c_char_p
        if built().decode() != source_hash():
            raise VoltHipError(f"{LIB} was built from other sources ({built().decode()} != {source_hash()}): "
                               "run `python -m volt_amd.build --force`")
        for name, (res, args) in _SIGS.items():
            fn = getattr(handle, name)           # AttributeError if the symbol is missing
            fn.restype, fn.argtypes = res, args
        if handle.volt_abi_version() != ABI_VERSION:
            raise VoltHipError("libvolt_hip.so ABI version mismatch: rebuild")
        _lib = handle
    return _lib


def check(rc: int, what: str) -> None:
    if rc == 0:
        return
    if rc < 0:
        raise VoltHipError(f"{what}: invalid argument #{-rc}")
    raise VoltHipError(f"{what}: HIP error {rc}")


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream
