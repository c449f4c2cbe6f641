"""ctypes binding of libvolt_hip.so (include/volt_hip.h).  No fallback: if the library is missing
or a call fails, this raises -- the product path never routes around the HIP kernels."""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  -- load torch's libamdhip64 first so the library binds the same HIP runtime

from .build import LIB, build_lib, source_hash, sources_present

_lib = None
ABI_VERSION = 5
WANT_GRAD, WS_INITIALISED, REFINE_ALPHA = 1, 2, 4       # include/volt_hip.h: VOLT_WANT_GRAD, VOLT_WS_INITIALISED, VOLT_REFINE_ALPHA

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
    "volt_potrf_workspace_init_f32": (C.c_int, [_ptr, C.c_size_t, _i32, _i32, _ptr]),
    "volt_potrf_ws_f32": (C.c_int, [_ptr, _ptr, _ptr, _i32, _i32, _ptr, C.c_size_t, _i32, _ptr]),
    "volt_potrf_k_f32": (C.c_int, [_ptr, _i64, _i64, _ptr, C.c_float, _ptr, _ptr, _ptr, _i32, _i32, _ptr, C.c_size_t, _i32, _ptr]),
    "volt_prepare_f64": (C.c_int, [_ptr, _i64, _i64, _ptr, _f64, _ptr, _i32, _i32, _ptr]),
    "volt_potrf_f64": (C.c_int, [_ptr, _ptr, _ptr, _i32, _i32, _ptr]),
    "volt_potrf_workspace_bytes_f64": (C.c_size_t, [_i32, _i32]),
    "volt_potrf_ws_f64": (C.c_int, [_ptr, _ptr, _ptr, _i32, _i32, _ptr, C.c_size_t, _ptr]),
    "volt_potrf_k_f64": (C.c_int, [_ptr, _i64, _i64, _ptr, C.c_double, _ptr, _ptr, _ptr, _i32, _i32, _ptr, C.c_size_t, _ptr]),
    "volt_trsv_lower_f64": (C.c_int, [_ptr, _ptr, _ptr, _ptr, _ptr, _i32, _i32, _ptr]),
    "volt_trsv_lower_t_f64": (C.c_int, [_ptr, _ptr, _ptr, _ptr, _ptr, _i32, _i32, _ptr]),
    "volt_trsv_lower_f32": (C.c_int, [_ptr, _ptr, _ptr, _ptr, _ptr, _i32, _i32, _ptr]),
    "volt_trsv_lower_t_f32": (C.c_int, [_ptr, _ptr, _ptr, _ptr, _ptr, _i32, _i32, _ptr]),
    "volt_trtri_f32": (C.c_int, [_ptr, _ptr, _ptr, _i32, _i32, _ptr]),
    "volt_trtri_f64": (C.c_int, [_ptr, _ptr, _ptr, _i32, _i32, _ptr]),
    "volt_trtri_workspace_bytes_f64": (C.c_size_t, [_i32, _i32]),
    "volt_trtri_ws_f64": (C.c_int, [_ptr, _ptr, _ptr, _i32, _i32, _ptr, C.c_size_t, _ptr]),
    "volt_mll_workspace_bytes_f64": (_sz, [_i32, _i32, _i32]),
    "volt_mll_step_f64": (C.c_int, [_ptr, _i64, _i64, _ptr, _ptr, _f64, _ptr, _ptr, _ptr, _ptr, _i32, _i32, _i32, _ptr]),
    "volt_rollout_scratch_bytes": (_sz, [_i32, _i32, _i32]),
    "volt_rollout_bordered_f32": (C.c_int, [_ptr] * 16 + [_i32] * 6 + [_f32] * 3 + [_ptr]),
    "volt_adam_step_f32": (C.c_int, [_ptr, _i32, C.c_longlong, _ptr, C.c_float, C.c_float, C.c_float, C.c_float, _ptr, _ptr]),
    "volt_mll_workspace_bytes": (_sz, [_i32, _i32, _i32]),
    "volt_mll_workspace_init_f32": (C.c_int, [_ptr, _i32, _i32, _i32, _ptr]),
    "volt_mll_step_f32": (C.c_int, [_ptr, _i64, _i64, _ptr, _ptr, _f32, _ptr, _ptr, _ptr, _ptr, _i32, _i32, _i32, _ptr]),
    "volt_mll_grad_k_f32": (C.c_int, [_ptr, _ptr, _ptr, _ptr, _i32, _i32, _ptr]),
    "volt_rollout_shared_f32": (C.c_int, [_ptr] * 8 + [_i32] * 5 + [_f32, _ptr]),
    "volt_gemm_nt_f32": (C.c_int, [_ptr, _i64, _i64, _i32, _ptr, _i64, _i64, _i32, _ptr, _i64, _i64, _i32, _f32, _f32,
                                   _i32, _i32, _i32, _i32, _ptr]),
    "volt_gpcv_workspace_bytes": (_sz, [_i32, _i32, _i32]),
    "volt_gpcv_step_f32": (C.c_int, [_ptr, _i64, _i64, _f32] + [_ptr] * 6 + [_i32, _f32, _f32, _f32, _f32] + [_ptr] * 7
                           + [_i32, _i32, _i32, _ptr]),
}

# measurement / tuning hooks: include/volt_hip_tune.h, not part of the drop-in boundary
_TUNE_SIGS = {
    "volt_profile_step_f32": (C.c_int, [_ptr, _i64, _i64, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _i32, _i32, _i32, _ptr, _ptr, _ptr,
                                        _ptr, _ptr]),
    "volt_tune_update_f32": (C.c_int, [_ptr, _ptr, _ptr, _i32, _i32, _i32, _i32, _i32, _ptr]),
    "volt_sched_describe": (C.c_int, [_i32, _i32, _i32, _i32, _i32, _i32, C.c_float, _ptr, _i32, _ptr]),
    "volt_tune_diag_f32": (C.c_int, [_ptr, _ptr, _ptr, _i32, _i32, _i32, _ptr, _ptr]),
    "volt_tune_diag_f64": (C.c_int, [_ptr, _ptr, _ptr, _i32, _i32, _i32, _ptr, _ptr]),
    "volt_tune_small_stamps": (C.c_int, [_ptr]),
    "volt_long_describe": (C.c_int, [_i32, _i32, _i32, _ptr, _i32, _ptr, _ptr]),
    "volt_batch_describe": (C.c_int, [_i32, _i32, _i32, _i32, _ptr, _i32]),
    "volt_tune_batch_stamps": (C.c_int, [_ptr]),
    "volt_batch64_describe": (C.c_int, [_i32, _i32, _i32, _ptr, _i32]),
    "volt_tune_batch64_stamps": (C.c_int, [_ptr]),
    "volt_topology_describe": (C.c_int, [_ptr]),
}

EXPORTS = tuple(_SIGS)                  # every symbol include/volt_hip.h declares
TUNE_EXPORTS = tuple(_TUNE_SIGS)        # every symbol include/volt_hip_tune.h declares


class VoltHipError(RuntimeError):
    pass


class VoltHipWarning(RuntimeWarning):
    """The library recovered from an internal condition (e.g. re-ran a step on the launch-per-column schedule)."""


# info[b] values at or below this are INTERNAL errors of the library, never "not positive definite" (include/volt_hip.h:
# INT_MIN = a hand-off inside a one-launch step timed out, INT_MIN + 1 = the workspace does not hold what its init wrote)
INFO_INTERNAL_MAX = -(2 ** 31) + 255


def _embedded_hash(path):
    handle = C.CDLL(path)
    fn = handle.volt_source_hash
    fn.restype = C.c_char_p
    return handle, fn().decode()


def lib() -> C.CDLL:
    """Load (once) and type the library.  A binary that is already there is dlopen'ed FIRST and judged by the source
    hash compiled into it: it is used when that equals the hash of the checked-in sources, or when the sources are not
    there to compare with (a deployed, read-only tree).  Only a missing or stale binary goes to hipcc (~30 s, behind a
    file lock).  Raises if neither works: volt_amd has no CPU fallback."""
    global _lib
    if _lib is None:
        handle = None
        if os.path.exists(LIB):
            try:
                handle, built = _embedded_hash(LIB)
            except (OSError, AttributeError):
                handle = None
            if handle is not None and sources_present() and built != source_hash():
                import _ctypes                                 # stale: built from other sources than the checked-in ones;
                _ctypes.dlclose(handle._handle)                # unmap it, or dlopen would hand the same image back after the rebuild
                handle = None
        if handle is None:
            try:
                build_lib(force=os.path.exists(LIB), verbose=False)
            except Exception as e:
                raise VoltHipError(
                    f"{LIB} is missing or stale and could not be built ({e}): run `python -m volt_amd.build` "
                    "(hipcc, gfx950). volt_amd has no CPU fallback.") from e
            handle, built = _embedded_hash(LIB)
            if built != source_hash():
                raise VoltHipError(f"{LIB} was built from other sources ({built} != {source_hash()}): "
                                   "run `python -m volt_amd.build --force`")
        for name, (res, args) in {**_SIGS, **_TUNE_SIGS}.items():
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
