"""CPU oracle for the GPCV volatility-extraction stage -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/`` and ``__graft_entry__.smoke()`` may import this module (same rule as
``oracle/volt_oracle.py``).  Plain torch on the CPU, any float dtype, differentiable by autograd.

SURVEY 8(f) row 4: ``LearnGPCV`` (``voltron/train_utils.py:15-67``), the first stage of both drivers
(``experiments/stocks/GenerateMultiMeanPreds.py``, ``experiments/weather/GPGenerator.py:64``).

What the reference builds there (citations under /root/reference):

* ``scaled_returns``                       ``voltron/train_utils.py:16-18``
* ``VolatilityGaussianLikelihood("exp")``  ``voltron/likelihoods/volatility_likelihood.py:42-50``:
  ``y_i | f_i ~ N(0, clamp(exp(f_i), min=1e-3))``
* ``SingleTaskVariationalGP(init_points=train_x, learn_inducing_locations=False,
  use_whitened_var_strat=False)``          ``voltron/models/single_task_variational_gp.py:69-122``:
  inducing points == training inputs, Cholesky variational distribution ``q(u) = N(m, Lq Lq')``,
  prior ``N(c 1, K)``, ``K`` = ``BMKernel`` (``voltron/kernels/BMKernel.py:38-45``) or ``FBMKernel``
  (``voltron/kernels/FBMKernel.py:38-45``)
* ``initialize_variational_parameters``    ``single_task_variational_gp.py:190-236``
* the loss ``-VariationalELBO(likelihood, model, N)(model(train_x), yy)`` under
  ``num_gauss_hermite_locs(75)``, Adam lr 0.01   ``train_utils.py:37-58``
* the readout ``likelihood(model(train_x)).scale.mean(0)``   ``train_utils.py:60-63``

PINNING STATUS: **reference-owned parts pinned; ELBO arithmetic unpinned.**  ``scaled_returns``, ``bm_cov``,
``init_variational`` (running std, clamped inverse Hessian, S = L (L'HL + I)^-1 L', 10 chol(S), the mean constant)
and the "exp" likelihood's scale are checked against ``tests/golden/gpcv.npz``, which
``tests/golden/make_golden_gpcv.py`` produces by EXECUTING the reference's own
``initialize_variational_parameters``, ``VolatilityGaussianLikelihood.forward`` and ``BMKernel.forward`` behind
dense stand-ins for the gpytorch names they import.  The arithmetic of the ELBO lives in gpytorch
(``UnwhitenedVariationalStrategy``, ``CholeskyVariationalDistribution``, ``VariationalELBO``,
``GaussHermiteQuadrature1D``, the MVN-MVN KL) and botorch (``GPyTorchModel``); neither is vendored
in /root/reference nor installed (setup.py:19 ``gpytorch>=1.0.1``, no pin), and the reference's two
files for this stage cannot be imported without them.  This restates the published algorithm:

* with inducing points equal to the inputs, the unwhitened strategy returns ``q(u)`` itself as the
  latent distribution (its ``torch.equal(x, inducing_points)`` short cut), and the prior is
  ``N(mean(Z), K(Z,Z) + 1e-3 I)`` (``add_jitter()`` default);
* ``expected_log_prob`` = Gauss-Hermite quadrature, nodes ``sqrt(2 var_i) x_k + m_i``, weights
  ``w_k / sqrt(pi)``, ``var_i = sum_j Lq[i,j]^2`` (clamped below at 1e-6, the MVN variance floor);
* ``ELBO = sum_i E_q[log p(y_i|f_i)] / N  -  KL(q(u) || p(u)) / N`` (``combine_terms``, beta = 1);
* ``KL = 1/2 (tr(K^-1 S) + (mu - m)' K^-1 (mu - m) - N + logdet K - logdet S)``.

It is checked in ``tests/test_oracle_gpcv.py`` against an independent dense fp64 evaluation through
``torch.distributions`` (MultivariateNormal KL + Normal log_prob on explicit quadrature nodes).
"""
from __future__ import annotations

import math

import numpy as np
import torch

PRIOR_JITTER = 1e-3          # LazyTensor.add_jitter() default, applied to the inducing prior covariance
MIN_VARIANCE = 1e-6          # gpytorch.settings.min_variance (fp32) floor in MultivariateNormal.variance
MIN_SCALE = 1e-3             # volatility_likelihood.py:50  .clamp(min=1e-3)
NUM_GH = 75                  # train_utils.py:50


def gauss_hermite(n: int = NUM_GH, dtype=torch.float64):
    """Nodes and weights of GaussHermiteQuadrature1D: numpy ``hermgauss(n)``; weights still to be / sqrt(pi)."""
    x, w = np.polynomial.hermite.hermgauss(n)
    return torch.as_tensor(x, dtype=dtype), torch.as_tensor(w, dtype=dtype)


def scaled_returns(train_x: torch.Tensor, train_y: torch.Tensor) -> torch.Tensor:
    """train_utils.py:16-18."""
    dt = train_x[1] - train_x[0]
    return (train_y[1:] - train_y[:-1]) / train_y[:-1] / (dt ** 0.5)


def bm_cov(x: torch.Tensor, vol: torch.Tensor) -> torch.Tensor:
    """BMKernel.forward, BMKernel.py:38-41: vol * min(x1, x2)."""
    return vol * torch.minimum(x.unsqueeze(-1), x.unsqueeze(-2))


def fbm_cov(x: torch.Tensor, vol: torch.Tensor) -> torch.Tensor:
    """FBMKernel.forward, FBMKernel.py:38-45 (``vol`` plays the Hurst exponent)."""
    a, b = x.unsqueeze(-1), x.unsqueeze(-2)
    h2 = 2.0 * vol
    return (a.abs().pow(h2) + b.abs().pow(h2) - (a - b).abs().pow(h2)) / 2.0


def psd_safe_cholesky(A: torch.Tensor, max_tries: int = 3):
    """gpytorch utils/cholesky.py: try as is, then add jitter 1e-6 (fp32) / 1e-8 (fp64) x 10^i to the diagonal."""
    L, info = torch.linalg.cholesky_ex(A)
    if not info.any():
        return L
    jitter = 1e-6 if A.dtype == torch.float32 else 1e-8
    Ap, prev = A.clone(), 0.0
    for i in range(max_tries):
        new = jitter * (10 ** i)
        Ap.diagonal(dim1=-2, dim2=-1).add_(new - prev)
        prev = new
        L, info = torch.linalg.cholesky_ex(Ap)
        if not info.any():
            return L
    raise RuntimeError("matrix not positive definite after jitter")


def running_std(y: torch.Tensor) -> torch.Tensor:
    """single_task_variational_gp.py:201-202: std of y[:i] (unbiased), the first ten set to entry 10."""
    N = y.shape[0]
    out = torch.full((N,), float("nan"), dtype=y.dtype)
    for i in range(2, N):
        out[i] = y[:i].std(0)
    out[:10] = out[10]
    return out


def init_variational(x: torch.Tensor, y: torch.Tensor, vol: float = 0.2, kernel: str = "bm"):
    """``initialize_variational_parameters`` for param == "exp" (single_task_variational_gp.py:190-236).

    Returns (variational_mean f, chol_variational_covar, mean constant).  Note the reference clamps the
    *whole* diag-embedded inverse Hessian to [1e-4, 1000], so its off-diagonal entries are 1e-4, not 0."""
    rs = running_std(y)
    f = rs.clamp(min=1e-4).log()
    ih = torch.diag_embed(0.5 * y.pow(-2.0) * (f * 2.0).exp()).clamp(min=1e-4, max=1000.0)
    volt = torch.as_tensor(vol, dtype=x.dtype)
    kuu = bm_cov(x, volt) if kernel == "bm" else fbm_cov(x, volt)
    L = psd_safe_cholesky(kuu)
    inner = L.mT @ ih @ L + torch.eye(x.shape[0], dtype=x.dtype)
    S = L @ torch.cholesky_solve(L.mT.contiguous(), psd_safe_cholesky(inner))
    S_root = psd_safe_cholesky(S).tril() * 10.0
    return f, S_root, rs.mean(0).log()


def elbo_terms(m, Lq_raw, const, K, y, gh_x, gh_w):
    """All pieces of the ELBO for one series (differentiable).  K is the prior covariance WITHOUT jitter."""
    N = y.shape[0]
    Lq = Lq_raw.tril()
    var = Lq.pow(2).sum(-1).clamp_min(MIN_VARIANCE)
    locs = torch.sqrt(2.0 * var).unsqueeze(0) * gh_x.unsqueeze(-1) + m.unsqueeze(0)          # [Q, N]
    scale = locs.exp().clamp(min=MIN_SCALE)
    logp = -((y.unsqueeze(0)) ** 2) / (2 * scale ** 2) - scale.log() - math.log(math.sqrt(2 * math.pi))
    ell = ((logp * gh_w.unsqueeze(-1)).sum(0) / math.sqrt(math.pi)).sum()
    Kj = K + PRIOR_JITTER * torch.eye(N, dtype=K.dtype)
    Lk = torch.linalg.cholesky(Kj)
    d = (m - const).unsqueeze(-1)
    T = torch.linalg.solve_triangular(Lk, Lq, upper=False)
    z = torch.linalg.solve_triangular(Lk, d, upper=False)
    trace, quad = T.pow(2).sum(), z.pow(2).sum()
    logdet_k = 2.0 * Lk.diagonal().log().sum()
    logdet_s = Lq.diagonal().pow(2).log().sum()
    kl = 0.5 * (trace + quad - N + logdet_k - logdet_s)
    return {"ell": ell, "kl": kl, "trace": trace, "quad": quad, "logdet_k": logdet_k, "logdet_s": logdet_s,
            "elbo": ell / N - kl / N}


def elbo(m, Lq_raw, const, raw_vol, x, y, kernel="bm", num_gh=NUM_GH):
    """ELBO (to be maximised; the reference's loss is its negative) as a function of the raw parameters."""
    gh_x, gh_w = gauss_hermite(num_gh, dtype=m.dtype)
    vol = torch.sigmoid(raw_vol)                                        # Interval(0, 1).transform
    K = bm_cov(x, vol) if kernel == "bm" else fbm_cov(x, vol)
    return elbo_terms(m, Lq_raw, const, K, y, gh_x, gh_w)["elbo"]


def elbo_and_grads(m, Lq_raw, const, raw_vol, x, y, kernel="bm", num_gh=NUM_GH):
    ps = [t.detach().clone().requires_grad_(True) for t in (m, Lq_raw, const, raw_vol)]
    val = elbo(*ps, x, y, kernel=kernel, num_gh=num_gh)
    grads = torch.autograd.grad(val, ps)
    return val.detach(), [g.detach() for g in grads]


def pred_scale(m, Lq_raw, eps):
    """train_utils.py:60-63: mean over the likelihood's function samples of clamp(exp(f), 1e-3), f = m + Lq eps.
    eps [n_samples, N] are the standard-normal draws (gpytorch draws 10, settings.num_likelihood_samples)."""
    f = m.unsqueeze(0) + eps @ Lq_raw.tril().mT
    return f.exp().clamp(min=MIN_SCALE).mean(0)


def learn_gpcv(train_x, train_y, train_iters=1000, kernel="bm", eps=None, dtype=torch.float32, record=None, init=None):
    """The whole of LearnGPCV on the CPU: init, Adam(lr=0.01) on -ELBO, readout.
    ``init`` = (variational mean, chol factor, mean constant) overrides the start-up values: the reference's start-up
    covariance has condition number > 1e6 (K[0,0] is pure jitter), so its smallest directions -- and with them
    logdet S -- are not reproducible between two fp32 LAPACKs; trajectory comparisons start from shared values."""
    x = train_x.to(dtype)
    yy = scaled_returns(train_x.to(dtype), train_y.to(dtype))
    if init is None:
        f, S_root, c0 = init_variational(x, yy, kernel=kernel)
    else:
        f, S_root, c0 = (t.to(dtype) for t in init)
    raw_vol0 = torch.logit(torch.tensor([0.2], dtype=dtype))
    ps = [f.clone().requires_grad_(True), S_root.clone().requires_grad_(True),
          c0.reshape(1).clone().requires_grad_(True), raw_vol0.clone().requires_grad_(True)]
    opt = torch.optim.Adam(ps, lr=0.01)
    for _ in range(train_iters):
        opt.zero_grad()
        loss = -elbo(ps[0], ps[1], ps[2], ps[3], x, yy, kernel=kernel)
        loss.backward()
        if record is not None:
            record.append(float(loss.detach()))
        opt.step()
    if eps is None:
        eps = torch.randn(10, x.shape[0], dtype=dtype)
    return pred_scale(ps[0].detach(), ps[1].detach(), eps.to(dtype)), [p.detach() for p in ps]
