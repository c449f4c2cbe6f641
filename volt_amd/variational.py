"""Variational-GP plumbing of the GPCV stage, backed by volt_gpcv_step_f32.

The reference builds this stage from gpytorch parts (voltron/train_utils.py:20-44,
voltron/models/single_task_variational_gp.py:69-122): ``CholeskyVariationalDistribution`` +
``UnwhitenedVariationalStrategy`` with the inducing points fixed at the training inputs, and
``VariationalELBO(likelihood, model, num_data, combine_terms=True)`` evaluated under
``num_gauss_hermite_locs(75)``.  gpytorch is third-party and absent from /root/reference; restated here
(from its published behaviour) is only what those call sites need:

* ``model(x)`` with ``x`` equal to the inducing points returns q(u) = N(m, Lq Lq') itself (the strategy's
  ``torch.equal(x, inducing_points)`` short cut); other inputs are outside the accelerated path;
* ``elbo = E_q[log p(y|f)].sum() / N - beta KL(q(u) || N(mean(Z), K(Z,Z) + 1e-3 I)) / num_data``;
* parameter names: ``variational_strategy._variational_distribution.{variational_mean,chol_variational_covar}``.

The arithmetic -- quadrature, Cholesky of the prior, the two triangular products behind tr(K^-1 S) and
K^-1 Lq, all gradients -- runs in libvolt_hip.so through one ``torch.autograd.Function``; no CPU path.
"""
from __future__ import annotations

import math

import numpy as np
import torch
from torch import nn

from . import gp, ops
from .gp import Module, MultivariateNormal, NotPSDError, NanError, _ScaledDense, _dense

PRIOR_JITTER = 1e-3        # LazyTensor.add_jitter() default on the inducing prior
MIN_VARIANCE = 1e-6        # gpytorch.settings.min_variance for fp32


class num_gauss_hermite_locs:
    """gpytorch.settings.num_gauss_hermite_locs (default 20): ``with num_gauss_hermite_locs(75): ...``."""
    _value = 20

    def __init__(self, value):
        self._new, self._old = int(value), None

    @classmethod
    def value(cls):
        return cls._value

    def __enter__(self):
        self._old, num_gauss_hermite_locs._value = num_gauss_hermite_locs._value, self._new
        return self

    def __exit__(self, *exc):
        num_gauss_hermite_locs._value = self._old
        return False


_GH_CACHE = {}


def _gauss_hermite(n, device):
    """GaussHermiteQuadrature1D's nodes and weights (numpy hermgauss), weights pre-divided by sqrt(pi)."""
    key = (n, str(device))
    if key not in _GH_CACHE:
        x, w = np.polynomial.hermite.hermgauss(n)
        _GH_CACHE[key] = (torch.tensor(x, dtype=torch.float32, device=device),
                          torch.tensor(w / math.sqrt(math.pi), dtype=torch.float32, device=device))
    return _GH_CACHE[key]


class CholeskyVariationalDistribution(Module):
    """q(u) = N(variational_mean, L L'), L = tril(chol_variational_covar); gpytorch initialises 0 / I."""

    def __init__(self, num_inducing_points, batch_shape=torch.Size(), **kwargs):
        super().__init__()
        self.variational_mean = nn.Parameter(torch.zeros(*batch_shape, num_inducing_points))
        self.chol_variational_covar = nn.Parameter(torch.eye(num_inducing_points).repeat(*batch_shape, 1, 1))


class UnwhitenedVariationalStrategy(Module):
    def __init__(self, model, inducing_points, variational_distribution, learn_inducing_locations=True):
        super().__init__()
        object.__setattr__(self, "model", model)
        if inducing_points.dim() == 1:
            inducing_points = inducing_points.unsqueeze(-1)
        if learn_inducing_locations:
            raise NotImplementedError("learn_inducing_locations=True is outside the accelerated path: LearnGPCV fixes "
                                      "the inducing points at the training inputs (train_utils.py:30)")
        self.register_buffer("inducing_points", inducing_points.detach().clone())
        self._variational_distribution = variational_distribution
        self.register_buffer("variational_params_initialized", torch.tensor(0))


class VariationalLatent(MultivariateNormal):
    """What ``model(train_x)`` returns: q(u) itself, tied to its model so the ELBO can reach the prior."""

    def __init__(self, model):
        dist = model.variational_strategy._variational_distribution
        self.model = model
        self.loc = dist.variational_mean
        self._chol = dist.chol_variational_covar

    @property
    def chol(self):
        return self._chol.tril()

    @property
    def _covar(self):
        L = self.chol.detach()
        return ops.gemm_nt(L, L, uplo_a=1, uplo_b=1)                 # S = L L'

    @property
    def variance(self):
        return self.chol.pow(2).sum(-1).clamp_min(MIN_VARIANCE)

    def rsample(self, sample_shape=torch.Size(), base_samples=None):
        """m + L eps, eps ~ N(0, I) of shape sample_shape + [N] (one series) -- CholLazyTensor's root is L itself."""
        if base_samples is None:
            base_samples = torch.randn(torch.Size(sample_shape) + self.loc.shape, dtype=self.loc.dtype,
                                       device=self.loc.device)
        eps = base_samples.reshape(-1, self.loc.shape[-1]) if self.loc.ndim == 1 else base_samples
        if self.loc.ndim == 1:
            f = ops.gemm_nt(eps, self.chol.detach(), uplo_b=1)       # [S,N] = eps L'
            return self.loc.detach() + f.reshape(base_samples.shape)
        S = eps.reshape(-1, *self.loc.shape)                          # [S,B,N]
        f = ops.gemm_nt(S.transpose(0, 1).contiguous(), self.chol.detach(), uplo_b=1)   # [B,S,N]
        return self.loc.detach() + f.transpose(0, 1).reshape(base_samples.shape)


class _GPCVElbo(torch.autograd.Function):
    """F[b] = w_ell ell_b - w_kl KL_b with the analytic gradient the HIP step returns."""

    @staticmethod
    def forward(ctx, m, Lq, mean, K, y, holder, scale, num_gh, w_ell, w_kl):
        B, n = m.shape
        want_dk = bool(ctx.needs_input_grad[3])
        gh_x, gh_w = _gauss_hermite(num_gh, m.device)
        ws = holder.workspace(B, n, want_dk, m.device)
        ops.gpcv_step(K.detach(), (m - mean).detach(), m.detach(), Lq.detach(), y, gh_x, gh_w, ws, want_dk=want_dk,
                      jitter=PRIOR_JITTER, min_var=MIN_VARIANCE, w_ell=w_ell, w_kl=w_kl)
        chk = gp.deferred_checks.deferring()
        if chk is None and gp.deferred_checks._active is not None:
            gp.deferred_checks._active.reserve(ws.info)
        if chk is not None:
            chk.note(ws.info)
        elif bool((ws.info != 0).any().item()):
            if ops.info_internal(ws.info):       # a hand-off time-out / workspace table: not a statement about K
                raise ops._lib.VoltHipError(f"volt_gpcv_step_f32: internal error, info = {ws.info.tolist()[:8]}")
            if torch.isnan(K).any() or torch.isnan(m).any() or torch.isnan(Lq).any():
                raise NanError("GPCV step: NaN in the prior covariance or the variational parameters")
            raise NotPSDError("GPCV step: prior covariance K + 1e-3 I is not positive definite")
        ctx.n, ctx.w_kl, ctx.has_scale = n, w_kl, scale is not None
        saved = [ws.grad_m.clone(), ws.grad_Lq.clone(), ws.grad_mu.clone()]
        if want_dk:
            saved.append(ws.grad_K.clone())
        if scale is not None:
            saved += [ws.out[:, 2:9].clone(), scale.detach().clone()]
        ctx.want_dk = want_dk
        ctx.save_for_backward(*saved)
        ctx.mark_non_differentiable()
        return ws.out[:, 9].clone()

    @staticmethod
    def backward(ctx, g):
        sv = list(ctx.saved_tensors)
        gm, gL, gmu = sv[:3]
        g1, g2 = g.reshape(-1, 1), g.reshape(-1, 1, 1)
        gK = g2 * sv[3] if ctx.want_dk else None
        gscale = None
        if ctx.has_scale:
            # K = c M + j I:  tr(K^-1 M) = (N - j tr K^-1)/c,  tr(G'MG) = (tr(K^-1 S) - j |G|^2)/c,
            #                 beta'M beta = (r'K^-1 r - j |beta|^2)/c;   dKL/dc = 1/2 (first - second - third)
            o, c = sv[-2], sv[-1]
            quad, tr_s, tr_inv, gg, bb = o[:, 0], o[:, 3], o[:, 4], o[:, 5], o[:, 6]
            j = PRIOR_JITTER
            dkl = 0.5 * ((ctx.n - j * tr_inv) - (tr_s - j * gg) - (quad - j * bb)) / c.reshape(-1)
            gscale = (-ctx.w_kl * g * dkl).reshape(c.shape) if c.numel() == g.numel() else \
                (-ctx.w_kl * g * dkl).sum().reshape(c.shape)
        return g1 * gm, g2 * gL, g1 * gmu, gK, None, None, gscale, None, None, None


class VariationalELBO(Module):
    """gpytorch.mlls.VariationalELBO(likelihood, model, num_data, beta=1.0, combine_terms=True) stand-in
    (train_utils.py:44): ``mll(model(train_x), yy)`` -> scalar ELBO (or [T] for a batched model)."""

    def __init__(self, likelihood, model, num_data, beta=1.0, combine_terms=True):
        super().__init__()
        if not combine_terms:
            raise NotImplementedError("combine_terms=False is not used on this path (train_utils.py:44)")
        if getattr(likelihood, "param", "exp") != "exp":
            raise NotImplementedError('only the "exp" volatility likelihood has an accelerated ELBO (train_utils.py:20)')
        object.__setattr__(self, "likelihood", likelihood)
        object.__setattr__(self, "model", model)
        self.num_data, self.beta = float(num_data), float(beta)
        self._ws = None

    def workspace(self, B, n, want_dk, device):
        ws = self._ws
        if ws is None or ws.B != B or ws.N != n or ws.want_dk != bool(want_dk) or ws.buf.device != device:
            self._ws = ws = ops.GpcvWorkspace(B, n, want_dk, device)
        return ws

    def forward(self, approximate_dist_f, target):
        if not isinstance(approximate_dist_f, VariationalLatent):
            raise TypeError("VariationalELBO expects the output of SingleTaskVariationalGP(train_x)")
        model = approximate_dist_f.model
        m, Lq = approximate_dist_f.loc, approximate_dist_f._chol
        if not m.is_cuda:
            raise ops._lib.VoltHipError("VariationalELBO: tensors must live on the MI355X; no CPU fallback")
        Z = model.variational_strategy.inducing_points
        prior = model.forward(Z)
        n = m.shape[-1]
        batched = m.ndim > 1
        m2, L3, y2 = m.reshape(-1, n), Lq.reshape(-1, n, n), target.reshape(-1, n).to(torch.float32)
        B = m2.shape[0]
        mean2 = prior.mean.expand(m.shape).reshape(-1, n)
        lazy = prior.lazy_covariance_matrix
        scale = None
        if isinstance(lazy, _ScaledDense):
            scale = lazy.scale
            K3 = (scale.detach().reshape(-1, 1, 1) * lazy.base).expand(B, n, n)
        else:
            K3 = _dense(lazy).expand(B, n, n) if _dense(lazy).ndim == 2 else _dense(lazy).reshape(-1, n, n)
        res = _GPCVElbo.apply(m2.to(torch.float32), L3.to(torch.float32), mean2.to(torch.float32), K3, y2, self, scale,
                              num_gauss_hermite_locs.value(), 1.0 / n, self.beta / self.num_data)
        return res.reshape(m.shape[:-1]) if batched else res.reshape(())
