"""Tensor-level wrappers over the C ABI (device pointers in, torch tensors out).

Everything here requires CUDA(HIP) tensors; there is deliberately no CPU path.  PyTorch is used
for device memory and streams only.
"""
from __future__ import annotations

import torch

from . import _lib

TILE = 128


def _need_gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.VoltHipError(
                "volt_amd ops need tensors on the MI355X (got a CPU tensor); there is no CPU fallback")


def padded_n(n: int) -> int:
    return (n + TILE - 1) // TILE * TILE


def cumtrapz(y: torch.Tensor, x: torch.Tensor, square: bool = False) -> torch.Tensor:
    """CumTrapz (voltron/kernels/VolKernel.py:4-10).  y [..., N]; x [N] or y-shaped."""
    _need_gpu(y, x)
    if y.dtype not in (torch.float32, torch.float64):
        y = y.float()
    x = x.to(y.dtype)
    n = y.shape[-1]
    if x.shape[-1] != n:
        raise ValueError("x and y disagree on N")
    yb = y.reshape(-1, n).contiguous()
    B = yb.shape[0]
    if x.ndim == 1:
        xb, bsx = x.contiguous(), 0
    else:
        xb = x.expand(y.shape).reshape(-1, n).contiguous()
        bsx = n
    V = torch.empty_like(yb)
    fn = _lib.lib().volt_cumtrapz_f32 if y.dtype == torch.float32 else _lib.lib().volt_cumtrapz_f64
    _lib.check(fn(yb.data_ptr(), n, xb.data_ptr(), bsx, V.data_ptr(), B, n, int(square), _lib.stream_ptr()),
               "volt_cumtrapz")
    return V.reshape(y.shape)


def fill(V: torch.Tensor) -> torch.Tensor:
    """K[..., i, j] = V[..., min(i, j)] (voltron/kernels/VolKernel.py:30-33)."""
    _need_gpu(V)
    n = V.shape[-1]
    Vb = V.reshape(-1, n).contiguous()
    B = Vb.shape[0]
    K = torch.empty(B, n, n, dtype=V.dtype, device=V.device)
    fn = _lib.lib().volt_fill_f32 if V.dtype == torch.float32 else _lib.lib().volt_fill_f64
    _lib.check(fn(Vb.data_ptr(), K.data_ptr(), B, n, n, n * n, _lib.stream_ptr()), "volt_fill")
    return K.reshape(*V.shape[:-1], n, n)


_TAPS = {}


def ewma_weights(k: int, device) -> torch.Tensor:
    """The reference's taps, computed with the same torch expression (voltron/means/EWMA.py:21-24); cached per
    (k, device) -- the reference rebuilds them (and a Conv1d) on every call, here that would be a host-to-device
    copy per training iteration."""
    key = (int(k), str(device))
    if key not in _TAPS:
        alpha = 2. / (k + 1)
        wghts = alpha * (1 - alpha) ** (torch.arange(k - 1, -1, -1))
        _TAPS[key] = (wghts / wghts.sum()).to(torch.float32).to(device)
    return _TAPS[key]


def ewma(y: torch.Tensor, k: int) -> torch.Tensor:
    """EWMA(y, k) (voltron/means/EWMA.py:20-37): y [..., N] -> [..., N+1] fp32, on the device."""
    _need_gpu(y)
    n = y.shape[-1]
    yb = y.to(torch.float32).reshape(-1, n).contiguous()
    B = yb.shape[0]
    w = ewma_weights(k, y.device)
    out = torch.empty(B, n + 1, dtype=torch.float32, device=y.device)
    _lib.check(_lib.lib().volt_ewma_f32(yb.data_ptr(), n, w.data_ptr(), k, out.data_ptr(), B, n, _lib.stream_ptr()),
               "volt_ewma")
    return out.reshape(*y.shape[:-1], n + 1)


def info_internal(info: torch.Tensor) -> int:
    """How many entries of a factorisation's ``info`` report an INTERNAL error of the library (a hand-off time-out inside a
    one-launch step, a workspace that does not hold its tables: include/volt_hip.h) -- as opposed to a non-positive pivot
    (info > 0), the only thing gpytorch's jitter ladder is for.  One host read."""
    return int((info <= _lib.INFO_INTERNAL_MAX).sum().item())


class CholeskyFactor:
    """Result of `potrf`: padded factor A [B,Np,Np] (lower), inverse diagonal blocks, info [B]."""

    def __init__(self, A, Winv, info, n):
        self.A, self.Winv, self.info, self.n = A, Winv, info, n

    @property
    def L(self) -> torch.Tensor:
        return torch.tril(self.A[:, : self.n, : self.n])


_POTRF_WS = {}


def _potrf_workspace(B: int, Np: int, device):
    """Caller-owned scratch of volt_potrf_ws_f32, one buffer per (device, stream, B, Np), reused across calls -- the
    jitter ladder and the many small-matrix call sites would otherwise allocate up to 277 MB per call.  Calls that
    share a buffer are ordered by their stream.  Returns (aligned pointer or None, bytes)."""
    nbytes = int(_lib.lib().volt_potrf_workspace_bytes(B, Np))
    if not nbytes:
        return None, 0
    key = (device.index, _lib.stream_ptr(), B, Np)
    buf = _POTRF_WS.get(key)
    if buf is None:
        if len(_POTRF_WS) >= 8:                                 # a handful of shapes is what a run has; drop the oldest
            _POTRF_WS.pop(next(iter(_POTRF_WS)))
        buf = _POTRF_WS[key] = torch.empty(nbytes + 256, dtype=torch.uint8, device=device)
        _lib.check(_lib.lib().volt_potrf_workspace_init_f32(((buf.data_ptr() + 255) // 256) * 256, nbytes, B, Np,
                                                            _lib.stream_ptr()), "volt_potrf_workspace_init")
    return ((buf.data_ptr() + 255) // 256) * 256, nbytes


def _potrf_ws_f64(B: int, Np: int, device):
    """The few KB of progress words the fp64 one-launch schedule wants (csrc/batch64_step.hip): one buffer per (device, stream,
    B, Np), as for the fp32 scratch.  Returns (aligned pointer or None, bytes)."""
    nbytes = int(_lib.lib().volt_potrf_workspace_bytes_f64(B, Np))
    if not nbytes:
        return None, 0
    key = (device.index, _lib.stream_ptr(), B, Np, "f64")
    buf = _POTRF_WS.get(key)
    if buf is None:
        if len(_POTRF_WS) >= 8:
            _POTRF_WS.pop(next(iter(_POTRF_WS)))
        buf = _POTRF_WS[key] = torch.empty(nbytes + 256, dtype=torch.uint8, device=device)
    return ((buf.data_ptr() + 255) // 256) * 256, nbytes


def potrf_f64_inplace(A: torch.Tensor, Winv: torch.Tensor, info: torch.Tensor) -> None:
    """volt_potrf_ws_f64 on a prepared [B,Np,Np] fp64 buffer (the factor replaces it)."""
    B, Np = A.shape[0], A.shape[1]
    wp, nbytes = _potrf_ws_f64(B, Np, A.device)
    _lib.check(_lib.lib().volt_potrf_ws_f64(A.data_ptr(), Winv.data_ptr(), info.data_ptr(), B, Np, wp, nbytes, _lib.stream_ptr()),
               "volt_potrf")


def potrf(K: torch.Tensor, sigma2: torch.Tensor | None = None, jitter: float = 0.0, tables: bool = True) -> CholeskyFactor:
    """Batched Cholesky of K + (sigma2 + jitter) I.  K [B,N,N] fp32 or fp64 (only the lower triangle is read); the
    factor keeps K's dtype (fp32: volt_potrf_f32 on v_mfma_f32_32x32x2; fp64: volt_potrf_f64 on v_mfma_f64_16x16x4).
    ``tables=False``: the launch-per-column schedules only (no scratch handed over, so no one-launch step) -- what the
    wrappers fall back to when a one-launch step reports an internal error (`info_internal`)."""
    _need_gpu(K, sigma2)
    if K.dtype not in (torch.float32, torch.float64) or K.ndim != 3:
        raise ValueError("potrf expects a [B,N,N] fp32 or fp64 tensor")
    if K.stride(-1) != 1:
        K = K.contiguous()
    B, n, _ = K.shape
    Np = padded_n(n)
    A = torch.empty(B, Np, Np, dtype=K.dtype, device=K.device)
    Winv = torch.empty(B, Np // TILE, TILE, TILE, dtype=K.dtype, device=K.device)
    info = torch.empty(B, dtype=torch.int32, device=K.device)
    s2 = None
    if sigma2 is not None:
        s2 = sigma2.to(K.dtype).expand(B).contiguous()
    L = _lib.lib()
    st = _lib.stream_ptr()
    s2p = s2.data_ptr() if s2 is not None else None
    if K.dtype == torch.float32:
        # straight from K (no copy-in pass); scratch for the small-batch schedules (0 bytes above 64 matrices and below
        # 3 block columns)
        wp, nbytes = _potrf_workspace(B, Np, K.device) if tables else (None, 0)
        _lib.check(L.volt_potrf_k_f32(K.data_ptr(), K.stride(1), K.stride(0), s2p, float(jitter), A.data_ptr(),
                                      Winv.data_ptr(), info.data_ptr(), B, n, wp, nbytes,
                                      _lib.WS_INITIALISED if wp else 0, st), "volt_potrf_k")
    else:
        # straight from K where the shape runs as one launch (small / medium batches); prepare + launch-per-column otherwise
        wp, nbytes = _potrf_ws_f64(B, Np, K.device) if tables else (None, 0)
        _lib.check(L.volt_potrf_k_f64(K.data_ptr(), K.stride(1), K.stride(0), s2p, float(jitter), A.data_ptr(), Winv.data_ptr(),
                                      info.data_ptr(), B, n, wp, nbytes, st), "volt_potrf_k")
    return CholeskyFactor(A, Winv, info, n)


def _pad_rhs(f: CholeskyFactor, rhs: torch.Tensor) -> torch.Tensor:
    B, Np = f.A.shape[0], f.A.shape[1]
    out = torch.zeros(B, Np, dtype=f.A.dtype, device=f.A.device)
    out[:, : f.n] = rhs.reshape(B, f.n)
    return out


def trsv(f: CholeskyFactor, rhs: torch.Tensor, transpose: bool = False, check: bool = False) -> torch.Tensor:
    """L^-1 rhs (or L^-T rhs) in the factor's dtype.  rhs [B,N] -> [B,N].  One launch (csrc/trsv.hip).
    ``check`` reads the solve's error word back (a device synchronisation) and raises if a hand-off timed out --
    without it a time-out is visible only as NaN in the result."""
    _need_gpu(rhs)
    r = _pad_rhs(f, rhs)
    out = torch.empty_like(r)
    scratch = torch.empty_like(r)
    B, Np = r.shape
    L = _lib.lib()
    if f.A.dtype == torch.float32:
        fn = L.volt_trsv_lower_t_f32 if transpose else L.volt_trsv_lower_f32
    else:
        fn = L.volt_trsv_lower_t_f64 if transpose else L.volt_trsv_lower_f64
    _lib.check(fn(f.A.data_ptr(), f.Winv.data_ptr(), r.data_ptr(), out.data_ptr(), scratch.data_ptr(), B, Np,
                  _lib.stream_ptr()), "volt_trsv")
    if check and int(scratch.view(torch.int32).reshape(-1)[1].item()) != 0:
        raise _lib.VoltHipError("volt_trsv: a block hand-off timed out (the result holds NaN blocks)")
    return out[:, : f.n]


def cholesky_solve(f: CholeskyFactor, rhs: torch.Tensor) -> torch.Tensor:
    """(L L^T)^-1 rhs for one right-hand side per matrix (torch.cholesky_solve, rollout_utils.py:36)."""
    return trsv(f, trsv(f, rhs), transpose=True)


def trtri(f: CholeskyFactor) -> torch.Tensor:
    """Y = L^-T as an upper-triangular [B,N,N] tensor in the factor's dtype (volt_trtri_f32 / volt_trtri_f64)."""
    B, Np = f.A.shape[0], f.A.shape[1]
    Y = torch.empty(B, Np, Np, dtype=f.A.dtype, device=f.A.device)
    L = _lib.lib()
    if f.A.dtype == torch.float32:
        _lib.check(L.volt_trtri_f32(f.A.data_ptr(), f.Winv.data_ptr(), Y.data_ptr(), B, Np, _lib.stream_ptr()), "volt_trtri")
    else:
        # a few KB of progress words let the whole inverse run as one launch (csrc/batch64_step.hip); 0 bytes: launch per row
        nbytes = int(L.volt_trtri_workspace_bytes_f64(B, Np))
        ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=f.A.device) if nbytes else None
        wp = (ws.data_ptr() + 255) // 256 * 256 if nbytes else None
        _lib.check(L.volt_trtri_ws_f64(f.A.data_ptr(), f.Winv.data_ptr(), Y.data_ptr(), B, Np, wp, nbytes, _lib.stream_ptr()), "volt_trtri")
    return torch.triu(Y[:, : f.n, : f.n])


class MllWorkspace:
    """Caller-owned scratch for volt_mll_step_f32 / _f64, reusable across steps of the same (B, N, dtype)."""

    def __init__(self, B: int, N: int, want_grad: bool, device, dtype=torch.float32):
        self.B, self.N, self.want_grad, self.dtype = B, N, bool(want_grad), dtype
        L = _lib.lib()
        query = L.volt_mll_workspace_bytes if dtype == torch.float32 else L.volt_mll_workspace_bytes_f64
        nbytes = query(B, N, int(want_grad))
        self.buf = torch.empty(nbytes + 256, dtype=torch.uint8, device=device)
        self.ptr = (self.buf.data_ptr() + 255) // 256 * 256
        self.flags = int(bool(want_grad)) * _lib.WANT_GRAD
        if dtype == torch.float32:                      # the launch-schedule table of this shape, once (mid-size batches)
            with torch.cuda.device(self.buf.device):
                _lib.check(L.volt_mll_workspace_init_f32(self.ptr, B, N, int(want_grad), _lib.stream_ptr()), "volt_mll_workspace_init")
            self.flags |= _lib.WS_INITIALISED           # ... and the step is told so (the library keeps no record of it)
        self.out = torch.empty(B, 8, dtype=dtype, device=device)
        self.alpha = torch.empty(B, N, dtype=dtype, device=device)
        self.info = torch.empty(B, dtype=torch.int32, device=device)

    def fits(self, B, N, want_grad, dtype=torch.float32):
        return self.B == B and self.N == N and self.want_grad == bool(want_grad) and self.dtype == dtype


def mll_step(K: torch.Tensor, resid: torch.Tensor, sigma2: torch.Tensor, ws: MllWorkspace | None = None,
             want_grad: bool = True, jitter: float = 0.0, refine_alpha: bool = False, tables: bool = True):
    """One MLL(+grad) evaluation with K resident, in K's dtype: fp32 -> volt_mll_step_f32 (v_mfma_f32_32x32x2), fp64 ->
    volt_mll_step_f64 (v_mfma_f64_16x16x4).  Returns (out [B,8], alpha [B,N], info [B]); see include/volt_hip.h for the
    meaning of out's columns.  ``refine_alpha`` (fp32, with want_grad; opt-in): one step of iterative refinement of alpha
    against K itself (VOLT_REFINE_ALPHA) -- both triangles of K must hold the symmetric matrix.
    ``tables=False`` (fp32): the workspace is NOT declared initialised, so the step runs the table-free launch-per-column
    schedules -- the fallback of the wrappers when a one-launch step reports an internal error (`info_internal`)."""
    _need_gpu(K, resid, sigma2)
    if K.ndim != 3 or K.dtype not in (torch.float32, torch.float64):
        raise ValueError("K must be [B,N,N] fp32 or fp64")
    dt = K.dtype
    B, n, _ = K.shape
    if K.stride(-1) != 1:
        K = K.contiguous()
    resid = resid.reshape(B, n).to(dt).contiguous()
    s2 = sigma2.to(dt).expand(B).contiguous()
    if ws is None or not ws.fits(B, n, want_grad, dt):
        ws = MllWorkspace(B, n, want_grad, K.device, dt)
    fn = _lib.lib().volt_mll_step_f32 if dt == torch.float32 else _lib.lib().volt_mll_step_f64
    _lib.check(fn(K.data_ptr(), K.stride(1), K.stride(0), resid.data_ptr(), s2.data_ptr(), float(jitter), ws.out.data_ptr(),
                  ws.alpha.data_ptr(), ws.info.data_ptr(), ws.ptr, B, n,
                  ((ws.flags if tables else ws.flags & ~_lib.WS_INITIALISED) | (_lib.REFINE_ALPHA if (refine_alpha and want_grad) else 0))
                  if dt == torch.float32 else int(want_grad),
                  _lib.stream_ptr()), "volt_mll_step")
    return ws.out, ws.alpha, ws.info


# ------------------------------------------------------------------ GPCV stage
def gemm_nt(A: torch.Tensor, B: torch.Tensor, uplo_a: int = 0, uplo_b: int = 0) -> torch.Tensor:
    """A @ B.mT on the library's fp32 MFMA GEMM.  A [T,M,K], B [T,N,K] (or 2-D); uplo_* = 1 / 2 declare an
    operand lower / upper triangular so its zero 128-blocks are skipped.  Operands are zero-padded to the
    128 tile here (plumbing); the arithmetic is volt_gemm_nt_f32."""
    _need_gpu(A, B)
    two_d = A.ndim == 2
    A3 = (A.unsqueeze(0) if two_d else A).to(torch.float32)
    B3 = (B.unsqueeze(0) if B.ndim == 2 else B).to(torch.float32)
    T, M, K = A3.shape
    N = B3.shape[1]
    if B3.shape[0] != T or B3.shape[2] != K:
        raise ValueError("gemm_nt: A [T,M,K] and B [T,N,K] disagree")
    Mp, Np_, Kp = padded_n(M), padded_n(N), padded_n(K)
    Ap = torch.zeros(T, Mp, Kp, dtype=torch.float32, device=A.device)
    Bp = torch.zeros(T, Np_, Kp, dtype=torch.float32, device=A.device)
    Ap[:, :M, :K] = A3
    Bp[:, :N, :K] = B3
    Cp = torch.empty(T, Mp, Np_, dtype=torch.float32, device=A.device)
    _lib.check(_lib.lib().volt_gemm_nt_f32(Ap.data_ptr(), Kp, Mp * Kp, uplo_a, Bp.data_ptr(), Kp, Np_ * Kp, uplo_b,
                                           Cp.data_ptr(), Np_, Mp * Np_, 0, 1.0, 0.0, T, Mp, Np_, Kp,
                                           _lib.stream_ptr()), "volt_gemm_nt")
    C = Cp[:, :M, :N]
    return C[0] if two_d else C


class GpcvWorkspace:
    """Caller-owned scratch and outputs of volt_gpcv_step_f32, reusable across steps of the same (B,N)."""

    def __init__(self, B: int, N: int, want_dk: bool, device):
        self.B, self.N, self.want_dk = B, N, bool(want_dk)
        nbytes = _lib.lib().volt_gpcv_workspace_bytes(B, N, int(want_dk))
        self.buf = torch.empty(nbytes + 256, dtype=torch.uint8, device=device)
        self.ptr = (self.buf.data_ptr() + 255) // 256 * 256
        with torch.cuda.device(self.buf.device):        # it begins with an MLL workspace: the schedule table of that step
            _lib.check(_lib.lib().volt_mll_workspace_init_f32(self.ptr, B, N, 1, _lib.stream_ptr()), "volt_mll_workspace_init")
        f32 = dict(dtype=torch.float32, device=device)
        self.out = torch.empty(B, 12, **f32)
        self.grad_m = torch.empty(B, N, **f32)
        self.grad_mu = torch.empty(B, N, **f32)
        self.grad_Lq = torch.empty(B, N, N, **f32)
        self.grad_K = torch.empty(B, N, N, **f32) if want_dk else None
        self.info = torch.empty(B, dtype=torch.int32, device=device)


def gpcv_step(K, resid, m, Lq, y, gh_x, gh_w, ws: GpcvWorkspace | None = None, want_dk: bool = False,
              jitter: float = 1e-3, min_var: float = 1e-6, min_scale: float = 1e-3, w_ell: float = 1.0,
              w_kl: float = 1.0):
    """One ELBO + gradient evaluation of the GPCV variational GP (include/volt_hip.h, volt_gpcv_step_f32).
    K [B,N,N] prior covariance without jitter; resid = m - prior mean, m, y [B,N]; Lq [B,N,N].
    Gradients are those of F = w_ell * ell - w_kl * KL (out[:, 9]).
    Returns the workspace: .out [B,12], .grad_m, .grad_mu, .grad_Lq (.grad_K if want_dk), .info."""
    _need_gpu(K, resid, m, Lq, y, gh_x, gh_w)
    if K.ndim != 3 or K.dtype != torch.float32:
        raise ValueError("K must be [B,N,N] fp32")
    B, n, _ = K.shape
    if K.stride(-1) != 1:
        K = K.contiguous()
    c = lambda t, shape: t.reshape(shape).to(torch.float32).contiguous()
    resid, m, y, Lq = c(resid, (B, n)), c(m, (B, n)), c(y, (B, n)), c(Lq, (B, n, n))
    gh_x, gh_w = gh_x.to(torch.float32).contiguous(), gh_w.to(torch.float32).contiguous()
    if ws is None or ws.B != B or ws.N != n or ws.want_dk != bool(want_dk):
        ws = GpcvWorkspace(B, n, want_dk, K.device)
    _lib.check(_lib.lib().volt_gpcv_step_f32(
        K.data_ptr(), K.stride(1), K.stride(0), float(jitter), resid.data_ptr(), m.data_ptr(), Lq.data_ptr(),
        y.data_ptr(), gh_x.data_ptr(), gh_w.data_ptr(), gh_x.numel(), float(min_var), float(min_scale), float(w_ell), float(w_kl),
        ws.out.data_ptr(), ws.grad_m.data_ptr(), ws.grad_mu.data_ptr(), ws.grad_Lq.data_ptr(),
        ws.grad_K.data_ptr() if want_dk else None, ws.info.data_ptr(), ws.ptr, B, n, _lib.WS_INITIALISED, _lib.stream_ptr()),
        "volt_gpcv_step")
    return ws


def mll_grad_k(ws: MllWorkspace) -> torch.Tensor:
    """d mll / d K = 1/2 (alpha alpha' - K_s^-1) / N  [B,N,N], from the Y = L^-T the last ``mll_step(want_grad=True)``
    left in ``ws`` (volt_mll_grad_k_f32) -- for kernels with trainable parameters."""
    if not ws.want_grad:
        raise ValueError("mll_grad_k needs a workspace used with want_grad=True")
    B, n = ws.B, ws.N
    Np = padded_n(n)
    scratch = torch.empty(B, Np, Np, dtype=torch.float32, device=ws.buf.device)
    gK = torch.empty(B, n, n, dtype=torch.float32, device=ws.buf.device)
    _lib.check(_lib.lib().volt_mll_grad_k_f32(ws.ptr, ws.alpha.data_ptr(), scratch.data_ptr(), gK.data_ptr(), B, n,
                                              _lib.stream_ptr()), "volt_mll_grad_k")
    return gK
