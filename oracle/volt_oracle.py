"""CPU oracle for Volt's exact-GP hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this module.  ``volt_amd`` never does: the product path
is the HIP library behind ``include/volt_hip.h`` and fails loudly without it.

This is a numpy restatement of the reference's algorithm for the path
(citations are ``file:line`` under ``/root/reference``):

=====================  =========================================================
function here          follows
=====================  =========================================================
``cumtrapz``           ``voltron/kernels/VolKernel.py:4-10``   (CumTrapz)
``volatility_kernel``  ``voltron/kernels/VolKernel.py:18-42``  (VolatilityKernel.forward)
``ewma`` + ``*_mean``  ``voltron/means/EWMA.py:20-37,39-54,74-91,94-113,116-135``
``generate_prediction````voltron/rollout_utils.py:6-53``
``rollouts``           ``voltron/rollout_utils.py:57-93``
``psd_safe_cholesky``  gpytorch ``utils/cholesky.py`` (NOT in /root/reference; restated
                       from the published algorithm, gpytorch>=1.0.1 per setup.py:19)
``noise_from_raw``     gpytorch ``GaussianLikelihood`` default ``GreaterThan(1e-4)``
                       softplus constraint (third-party, restated)
``mll_and_grads``      gpytorch ``ExactMarginalLogLikelihood`` / ``MultivariateNormal.log_prob``
                       call sites ``voltron/train_utils.py:127,240,249``; closed form
``nonvol_rollouts``    ``voltron/rollout_utils.py:95-115``; ``gp_posterior`` = the latent exact-GP
                       predictive behind botorch's ``model.posterior`` (third-party, restated)
=====================  =========================================================

PINNING STATUS
--------------
* ``cumtrapz``, ``volatility_kernel``, ``ewma`` family, ``generate_prediction`` and
  ``rollouts`` are PINNED: ``tests/golden/make_golden.py`` executes the reference's own
  files (loaded one by one from /root/reference with stand-ins for three gpytorch names)
  and commits their outputs under ``tests/golden/``; ``tests/test_oracle_golden.py``
  checks this module against them.
* ``mll_and_grads`` is **parity unpinned**: the arithmetic lives in
  gpytorch, which is not vendored in /root/reference and is not installed.  It is checked
  against fp64 autograd of the dense Gaussian log-density and against the analytic
  identities of this kernel (chol(K)[i,j] = sqrt(d_j)), never against gpytorch itself.
* ``psd_safe_cholesky`` is a restatement as well; vectors that pass through it are
  pinned to LAPACK potrf + this jitter loop.

Number types: the reference computes in fp32 (EWMA even forces FloatTensor).  Functions
here keep the input dtype where the reference does, and mimic torch's CPU semantics where
they are observable bit-for-bit: ``torch.cumsum`` on fp32 CPU tensors accumulates in
double and rounds every prefix to fp32 (verified in make_golden.py).
"""
from __future__ import annotations

import math
import numpy as np
import scipy.linalg as sla

LOG_2PI = math.log(2.0 * math.pi)


# --------------------------------------------------------------------------- a1
def cumtrapz(y: np.ndarray, x: np.ndarray) -> np.ndarray:
    """VolKernel.py:4-10.  dx = x[1]-x[0] (uniform grid assumed), weights dx with the
    FIRST and LAST halved, then a plain cumulative sum along the last axis."""
    y = np.asarray(y)
    x = np.asarray(x)
    dt = np.result_type(y.dtype, x.dtype)
    dx = x[..., 1] - x[..., 0]
    if x.ndim > 1:
        dx = dx[..., None]
    w = (dx * np.ones_like(x)).astype(x.dtype)
    w[..., 0] *= x.dtype.type(0.5)
    w[..., -1] *= x.dtype.type(0.5)
    prod = (w * y).astype(dt)
    # torch CPU cumsum: accumulate in double, round each prefix to the tensor dtype.
    return np.cumsum(prod.astype(np.float64), axis=-1).astype(dt)


# --------------------------------------------------------------------------- a2
def volatility_kernel_last_dim_is_batch(x: np.ndarray, vol_path: np.ndarray, diag: bool = False) -> np.ndarray:
    """VolKernel.py:24-26,35-40 (``last_dim_is_batch=True``; "TODO: check this" in the reference, mirrored as written):
    vol_path [N, D] is transposed, K [D,N,N] permuted to [N,N,D]; with ``diag`` the diagonal of THAT over its last two
    dims, i.e. out[i, d] = V[d, min(i, d)] for d < min(N, D)."""
    K = volatility_kernel(x, np.swapaxes(np.asarray(vol_path), -1, -2))          # [D,N,N]
    res = np.transpose(K, (1, 2, 0))
    return np.diagonal(res, axis1=-2, axis2=-1).copy() if diag else res


def volatility_kernel(x: np.ndarray, vol_path: np.ndarray, diag: bool = False) -> np.ndarray:
    """VolKernel.py:18-42.  K[..., i, j] = V[..., min(i, j)], V = cumtrapz(vol^2, x).
    The second argument is the volatility path, not a second set of inputs."""
    x = np.asarray(x)
    vol_path = np.asarray(vol_path)
    if x.shape[-1] == 1:
        x = np.squeeze(x)
    if vol_path.shape[-1] == 1:
        vol_path = np.squeeze(vol_path)
    vol_int = cumtrapz(vol_path * vol_path, x)
    n = x.shape[-1]
    idx = np.minimum.outer(np.arange(n), np.arange(n))
    res = vol_int[..., idx]
    if diag:
        return np.diagonal(res, axis1=-2, axis2=-1)
    return res


# --------------------------------------------------------------------------- a4
def ewma_weights(k: int) -> np.ndarray:
    """EWMA.py:21-24: w_j ~ alpha (1-alpha)^(k-1-j), normalised; fp32 like torch."""
    alpha = 2.0 / (k + 1)
    w = np.float32(alpha) * np.power(np.float32(1.0 - alpha), np.arange(k - 1, -1, -1)).astype(np.float32)
    return (w / w.sum(dtype=np.float32)).astype(np.float32)


def ewma(y: np.ndarray, k: int) -> np.ndarray:
    """EWMA.py:20-37.  Left-pad with k copies of y[..., 0], valid correlation with the
    k weights: out[..., t] = sum_j w_j * padded[..., t + j],  t = 0..N  (length N+1).
    out[t] only sees y[t-k .. t-1]; out[N] is the one-step-ahead value."""
    y = np.asarray(y, dtype=np.float32)
    w = ewma_weights(k)
    pad = np.repeat(y[..., :1], k, axis=-1)
    padded = np.concatenate([pad, y], axis=-1)
    n = y.shape[-1]
    out = np.zeros(y.shape[:-1] + (n + 1,), dtype=np.float64)
    for j in range(k):
        out += np.float64(w[j]) * padded[..., j:j + n + 1].astype(np.float64)
    return out.astype(np.float32)


def _select(ma: np.ndarray, x: np.ndarray, train_x: np.ndarray) -> np.ndarray:
    """The three-way return shared by every mean in EWMA.py (e.g. :46-54)."""
    x = np.asarray(x)
    if x.size == 1:
        return ma[..., -1][None, ...]
    if np.array_equal(np.squeeze(x), np.squeeze(train_x)):
        return ma[..., :-1]
    return ma


def ewma_mean(x, train_x, train_y, k=20):
    """EWMAMean.forward, EWMA.py:46-54."""
    return _select(ewma(train_y, k), x, train_x)


def dewma_mean(x, train_x, train_y, k=20):
    """DEWMAMean.forward, EWMA.py:81-91: 2*ema - ema(ema)[:-1]."""
    ema = ewma(train_y, k)
    ema_ema = ewma(ema, k)[..., :-1]
    return _select((2 * ema - ema_ema).astype(np.float32), x, train_x)


def tewma_mean(x, train_x, train_y, k=20):
    """TEWMAMean.forward, EWMA.py:102-113: 3*ema - 3*ema2 + ema3."""
    ema = ewma(train_y, k)
    ema2 = ewma(ema, k)[..., :-1]
    ema3 = ewma(ema2, k)[..., :-1]
    return _select((3 * ema - 3 * ema2 + ema3).astype(np.float32), x, train_x)


def meanrevert_mean(x, train_x, train_y, k=20, theta=0.5, latent=None):
    """MeanRevertingEMAMean.forward, EWMA.py:126-135.  ``latent`` is the mean of train_y
    taken at CONSTRUCTION time (:124); Rollouts later mutates train_y but not it."""
    train_y = np.asarray(train_y, dtype=np.float32)
    if latent is None:
        latent = train_y.mean(dtype=np.float32)
    latent = np.float32(latent)
    ema = ewma(train_y, k).copy()
    ema[..., 1:] -= np.float32(theta) * (ema[..., :-1] - latent)
    return _select(ema, x, train_x)


# ------------------------------------------------------------- third-party restated
def softplus(x):
    x = np.asarray(x, dtype=np.float64)
    return np.where(x > 20.0, x, np.log1p(np.exp(np.minimum(x, 20.0))))


def noise_from_raw(raw_noise, lower=1e-4):
    """gpytorch GaussianLikelihood: noise = softplus(raw_noise) + 1e-4 (GreaterThan(1e-4)).
    With the reference's raw_noise = 1e-5 (train_utils.py:222) sigma^2 ~= 0.6933."""
    return softplus(raw_noise) + lower


def psd_safe_cholesky(a: np.ndarray, jitter=None, max_tries: int = 3):
    """gpytorch.utils.cholesky.psd_safe_cholesky restated: try potrf; on failure add
    jitter * 10^i to the diagonal for i = 0..max_tries-1 (jitter default 1e-6 fp32 /
    1e-8 fp64) and retry; raise if still not PD.  Returns (L, jitter_used)."""
    a = np.asarray(a)
    if np.isnan(a).any():
        raise FloatingPointError("cholesky: NaN in input")
    try:
        return np.linalg.cholesky(a), 0.0
    except np.linalg.LinAlgError:
        pass
    if jitter is None:
        jitter = 1e-6 if a.dtype == np.float32 else 1e-8
    eye = np.eye(a.shape[-1], dtype=a.dtype)
    prev = 0.0
    for i in range(max_tries):
        new = jitter * (10 ** i)
        a = a + a.dtype.type(new - prev) * eye
        prev = new
        try:
            return np.linalg.cholesky(a), new
        except np.linalg.LinAlgError:
            continue
    raise np.linalg.LinAlgError(f"matrix not positive definite after jitter {prev:g}")


# --------------------------------------------------------------------------- a5
def mll_and_grads(K: np.ndarray, y: np.ndarray, mean: np.ndarray, raw_noise: float):
    """Exact-GP marginal log likelihood per datum and its analytic gradient (fp64).

        sigma2 = softplus(raw) + 1e-4 ;  Ks = K + sigma2 I ;  r = y - mean
        mll    = -1/2 ( r' Ks^-1 r + log|Ks| + N log 2pi ) / N
        d mll / d sigma2 = 1/2 ( a'a - tr Ks^-1 ) / N ,  a = Ks^-1 r
        d sigma2 / d raw = sigmoid(raw)
        d mll / d mean   = a / N

    Call sites: train_utils.py:249 (loss = -mll(output, y)), :250 (backward).
    PARITY UNPINNED (gpytorch absent): see module docstring.  K may be [N,N] or [B,N,N].
    Returns dict of fp64 arrays: mll, sigma2, quad, logdet, trinv, aa, d_raw, d_mean, alpha.
    """
    K = np.asarray(K, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    mean = np.asarray(mean, dtype=np.float64)
    if K.ndim == 2:
        out = mll_and_grads(K[None], y[None], mean[None], np.asarray(raw_noise).reshape(1))
        return {k: v[0] for k, v in out.items()}
    B, n, _ = K.shape
    raw = np.broadcast_to(np.asarray(raw_noise, dtype=np.float64).reshape(-1), (B,)).copy()
    sig2 = noise_from_raw(raw)
    res = {k: np.zeros(B) for k in ("mll", "quad", "logdet", "trinv", "aa", "d_raw")}
    res["sigma2"] = sig2
    res["alpha"] = np.zeros((B, n))
    res["d_mean"] = np.zeros((B, n))
    for b in range(B):
        Ks = K[b] + sig2[b] * np.eye(n)
        L = np.linalg.cholesky(Ks)
        r = y[b] - mean[b]
        z = sla.solve_triangular(L, r, lower=True)
        a = sla.solve_triangular(L, z, lower=True, trans="T")
        Linv = sla.solve_triangular(L, np.eye(n), lower=True)
        quad = float(z @ z)
        logdet = 2.0 * float(np.log(np.diag(L)).sum())
        trinv = float((Linv * Linv).sum())
        aa = float(a @ a)
        res["quad"][b] = quad
        res["logdet"][b] = logdet
        res["trinv"][b] = trinv
        res["aa"][b] = aa
        res["mll"][b] = -0.5 * (quad + logdet + n * LOG_2PI) / n
        dsig = 0.5 * (aa - trinv) / n
        res["d_raw"][b] = dsig / (1.0 + math.exp(-raw[b]))
        res["alpha"][b] = a
        res["d_mean"][b] = a / n
    return res


# --------------------------------------------------------------------------- a7
def generate_prediction(model_train_x, model_train_y, log_vol_path, test_x, pred_vol, z,
                        mean_fn, latent_mean=None, theta=0.5, jitter=1e-4):
    """rollout_utils.py:6-53 with the N(0,1) draw ``z`` [S,1,1] passed in instead of
    ``torch.randn`` (:47) so paths are comparable draw by draw.

    model_train_x [N'] or [S,N'], model_train_y [N'] or [S,N'] (log prices), log_vol_path
    [N'] or [S,N'] -- the *model attributes* GeneratePrediction reads (:7,:17,:32);
    test_x [T]; pred_vol [S,T]; z [S,T,1]; mean_fn(x) -> the model's mean module (:31,:39).
    Everything in fp32 like the reference; Cholesky via LAPACK spotrf.
    Returns samples [S, T] (matches ``(samples + pred_mean).squeeze(-1)``, :53).
    """
    f32 = np.float32
    train_x = np.asarray(model_train_x, dtype=f32)
    train_y = np.asarray(model_train_y, dtype=f32)
    test_x = np.asarray(test_x, dtype=f32)
    pred_vol = np.asarray(pred_vol, dtype=f32)
    vol = np.exp(np.asarray(log_vol_path, dtype=f32))
    S = pred_vol.shape[0]
    if train_x.ndim != test_x.ndim:                              # :8-11
        test_x_for_stack = np.repeat(test_x[None, :], train_x.shape[0], axis=0)
    else:
        test_x_for_stack = test_x
    vol_for_stack = np.repeat(vol[None, :], S, axis=0) if vol.ndim == 1 else vol   # :12-15
    full_x = np.concatenate([train_x, test_x_for_stack], axis=-1)                  # :17
    full_vol = np.concatenate([vol_for_stack, pred_vol], axis=-1)                  # :20
    idx_cut = train_x.shape[-1]                                                    # :24
    cov = volatility_kernel(full_x, full_vol)                                      # :26
    K_tr = cov[..., :idx_cut, :idx_cut]
    K_tr_te = cov[..., :idx_cut, idx_cut:]
    K_te = cov[..., idx_cut:, idx_cut:]
    train_mean = np.asarray(mean_fn(train_x), dtype=f32)                           # :31
    train_diffs = (train_y - train_mean)[..., None]                                # :32
    if train_diffs.ndim == 2:
        train_diffs = np.broadcast_to(train_diffs, (S,) + train_diffs.shape)
    T = test_x.shape[0]
    out = np.zeros((S, T), dtype=f32)
    tm = np.asarray(mean_fn(test_x), dtype=f32)                                    # :39
    tm = tm.T[..., None]                                   # .T.unsqueeze(-1): [T,1] / [1,1] / [S,1,1]
    tm = np.broadcast_to(tm, (S, T, 1))
    z = np.asarray(z, dtype=f32).reshape(S, T, 1)
    for s in range(S):
        L, _ = psd_safe_cholesky(K_tr[s].astype(f32), jitter=jitter)              # :35
        L = L.astype(f32)
        sol = sla.cho_solve((L, True), train_diffs[s].astype(f32)).astype(f32)     # :36
        pm = (K_tr_te[s].T @ sol).astype(f32) + tm[s]                              # :36,:39
        if latent_mean is not None:                                                # :41-42
            pm = pm - f32(theta) * (pm - f32(latent_mean))
        sol2 = sla.cho_solve((L, True), K_tr_te[s].astype(f32)).astype(f32)        # :44
        pc = (K_te[s] - K_tr_te[s].T @ sol2).astype(f32)
        pcL, _ = psd_safe_cholesky(pc, jitter=jitter)                              # :46
        out[s, :] = (pcL.astype(f32) @ z[s] + pm)[:, 0]                            # :48,:53
    return out


# --------------------------------------------------------------------------- a8
def rollouts(train_x, train_y, test_x, log_vol_path, pred_vol, z, mean_name="ewma", k=25,
             theta=None, mean_theta=0.5):
    """rollout_utils.py:57-93 (method == "volt") with ``pred_vol`` [S,H] (:66, out of scope:
    a sample of the BM vol model) and the normal draws ``z`` [S,H] passed in.

    train_x [N], train_y [N+1] RAW prices, test_x [H], log_vol_path [N]: the model state at
    entry (model.train_x = train_x, model.train_y = log(train_y[1:]), mean module built on
    those).  Reproduces the in-place mutation of the model between steps (:80-86) by
    rebuilding the state each step.  Returns samples [S,H] fp32 (log-price units).
    """
    f32 = np.float32
    train_x = np.asarray(train_x, dtype=f32)
    train_y = np.asarray(train_y, dtype=f32)
    test_x = np.asarray(test_x, dtype=f32)
    pred_vol = np.asarray(pred_vol, dtype=f32)
    z = np.asarray(z, dtype=f32)
    S, H = pred_vol.shape
    latent_mean = None if theta is None else np.log(train_y).mean(dtype=f32)       # :60-63
    means = {"ewma": lambda x, tx, ty: ewma_mean(x, tx, ty, k),
             "dewma": lambda x, tx, ty: dewma_mean(x, tx, ty, k),
             "tewma": lambda x, tx, ty: tewma_mean(x, tx, ty, k),
             "meanrevert": lambda x, tx, ty: meanrevert_mean(x, tx, ty, k, mean_theta, mr_latent)}
    mfn = means[mean_name]
    samples = np.zeros((S, H), dtype=f32)
    log_y = np.log(train_y[1:]).astype(f32)
    mr_latent = log_y.mean(dtype=f32)
    m_tx, m_ty, m_lv = train_x, log_y, np.asarray(log_vol_path, dtype=f32)
    samples[:, 0] = generate_prediction(m_tx, m_ty, m_lv, test_x[0:1], pred_vol[:, 0:1], z[:, 0],
                                        lambda x: mfn(x, m_tx, m_ty), latent_mean,
                                        0.5 if theta is None else theta)[:, 0]     # :67-70
    stack_y0 = np.repeat(log_y[None, :], S, axis=0)                                # :71
    stack_v0 = np.repeat(m_lv[None, :], S, axis=0)                                 # :72
    for idx in range(1, H):                                                        # :74
        stack_y = np.concatenate([stack_y0, samples[:, :idx]], axis=-1)
        stack_vol = np.concatenate([stack_v0, np.log(pred_vol[:, :idx])], axis=-1)
        rolling_x = np.concatenate([train_x, test_x[:idx]])
        samples[:, idx] = generate_prediction(
            rolling_x, stack_y, stack_vol, test_x[idx:idx + 1], pred_vol[:, idx:idx + 1], z[:, idx],
            lambda x, rx=rolling_x, sy=stack_y: mfn(x, rx, sy), latent_mean,
            0.5 if theta is None else theta)[:, 0]                                 # :87-90
    return samples


# ------------------------------------------------------------------ (f)2: baseline GPs
def rbf_kernel(a, b, lengthscale, outputscale=1.0):
    """gpytorch ScaleKernel(RBFKernel()): os * exp(-1/2 (a-b)^2 / ls^2)   (BasicWind.py:30-34)."""
    d = (np.asarray(a, dtype=np.float64)[:, None] - np.asarray(b, dtype=np.float64)[None, :]) / lengthscale
    return outputscale * np.exp(-0.5 * d * d)


def matern_kernel(a, b, lengthscale, outputscale=1.0, nu=2.5):
    """gpytorch ScaleKernel(MaternKernel(nu)) (BasicGPModels.py:10): os * c_nu(d) exp(-sqrt(2 nu) d), d = |a-b|/ls."""
    d = np.abs(np.asarray(a, dtype=np.float64)[:, None] - np.asarray(b, dtype=np.float64)[None, :]) / lengthscale
    c = {0.5: 1.0, 1.5: 1.0 + math.sqrt(3) * d, 2.5: 1.0 + math.sqrt(5) * d + 5.0 / 3.0 * d * d}[nu]
    return outputscale * c * np.exp(-math.sqrt(2 * nu) * d)


def sm_kernel(a, b, weights, means, scales):
    """gpytorch SpectralMixtureKernel, 1-D inputs: sum_q w_q exp(-2 pi^2 tau^2 v_q^2) cos(2 pi tau mu_q)."""
    tau = np.asarray(a, dtype=np.float64)[:, None] - np.asarray(b, dtype=np.float64)[None, :]
    out = np.zeros_like(tau)
    for w, mu, v in zip(weights, means, scales):
        out += w * np.exp(-2 * math.pi ** 2 * tau ** 2 * v ** 2) * np.cos(2 * math.pi * tau * mu)
    return out


def gp_posterior(kfun, noise, xt, y, mean_t, xs, mean_s):
    """Latent exact-GP predictive (botorch ``posterior``, observation_noise=False):  y [.., N] may be stacked."""
    Ktt = kfun(xt, xt) + noise * np.eye(len(xt))
    Kst = kfun(xs, xt)
    sol = np.linalg.solve(Ktt, (np.asarray(y, dtype=np.float64) - mean_t)[..., None])[..., 0]
    mean = mean_s + sol @ Kst.T
    cov = kfun(xs, xs) - Kst @ np.linalg.solve(Ktt, Kst.T)
    return mean, cov


def nonvol_rollouts(train_x, train_y, test_x, kfun, noise, z, mean_name="ewma", k=20, mean_theta=0.5):
    """rollout_utils.py:95-115 with the normal draws ``z`` [S,H] passed in.  train_x [N], train_y [N] RAW prices
    (the reference stacks ``train_y.log()`` whole, :100), kfun(a, b) the model's covariance, noise the likelihood's.
    Step 0 samples the un-stacked model S times (:99); every later step re-conditions the S stacked series (:102-114)
    from scratch, exactly as the reference does."""
    f32 = np.float32
    train_x = np.asarray(train_x, dtype=f32)
    log_y = np.log(np.asarray(train_y, dtype=f32)).astype(f32)
    test_x = np.asarray(test_x, dtype=f32)
    z = np.asarray(z, dtype=np.float64)
    S, H = z.shape
    mr_latent = log_y.mean(dtype=f32)                         # MeanRevertingEMAMean fixes it at construction (EWMA.py:124)

    def mfull(y):                                             # the means' third branch: train points + next point
        e1 = ewma(y, k)
        if mean_name == "ewma":
            return e1
        if mean_name == "meanrevert":                         # EWMA.py:126-128
            e = e1.copy()
            e[..., 1:] -= f32(mean_theta) * (e1[..., :-1] - mr_latent)
            return e
        e2 = ewma(e1, k)[..., :-1]                            # EWMA.py:83-84
        if mean_name == "dewma":
            return 2 * e1 - e2
        e3 = ewma(e2, k)[..., :-1]                            # EWMA.py:104-106
        return 3 * e1 - 3 * e2 + e3
    samples = np.zeros((S, H), dtype=f32)
    stack0 = np.repeat(log_y[None, :], S, axis=0)
    for idx in range(H):
        stack_y = log_y if idx == 0 else np.concatenate([stack0, samples[:, :idx]], axis=-1)
        rolling_x = np.concatenate([train_x, test_x[:idx]])
        ma = mfull(stack_y).astype(np.float64)                # [.., N+idx+1]: train-point means then the next point's
        mean, cov = gp_posterior(kfun, noise, rolling_x, stack_y, ma[..., :-1], test_x[idx:idx + 1], ma[..., -1:])
        sd = math.sqrt(max(float(cov[0, 0]), 0.0))
        samples[:, idx] = (np.broadcast_to(mean, (S, 1))[:, 0] if idx == 0 else mean[:, 0]) + sd * z[:, idx]
    return samples


# ------------------------------------------------------------------ analytic identities
def analytic_cholesky(V: np.ndarray) -> np.ndarray:
    """Known-answer structure: K = C diag(d) C', d = increments of V, so
    chol(K)[i, j] = sqrt(d_j) for i >= j.  For tests only; never a product shortcut."""
    V = np.asarray(V, dtype=np.float64)
    d = np.diff(np.concatenate([[0.0], V]))
    n = V.shape[0]
    return np.tril(np.broadcast_to(np.sqrt(d)[None, :], (n, n)))
