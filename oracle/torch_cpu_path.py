"""Torch-CPU restatement of the reference's "MLL + grad" training step -- TEST
INFRASTRUCTURE / CPU BASELINE ONLY (see oracle/volt_oracle.py header for who may import).

The reference's step is ``output = model(train_x); loss = -mll(output, y); loss.backward()``
(voltron/train_utils.py:243-250, :130-139; voltron/models/Volt.py:133-146).  Its arithmetic
runs inside gpytorch (ExactMarginalLogLikelihood -> GaussianLikelihood adds sigma^2 I ->
MultivariateNormal.log_prob -> Cholesky path for N <= max_cholesky_size), which is NOT in
/root/reference and not installed: this file restates that dense path with the same ATen ops
(``linalg.cholesky`` -> triangular solve -> log-diag, autograd backward through them) so it can
be timed on the host cores next to the HIP path.  PARITY UNPINNED against gpytorch itself.
K is the cached ``train_cov`` (VoltMagpie.py:46,123-124), so the fill is not part of a step.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

LOG_2PI = math.log(2.0 * math.pi)


def mll_step(K: torch.Tensor, y: torch.Tensor, mean: torch.Tensor, raw_noise: torch.Tensor):
    """One forward+backward.  K [B,N,N] (or [N,N]), y/mean [B,N], raw_noise [B] or [1]
    (requires_grad).  Returns (mll [B] detached, d mll / d raw_noise)."""
    if K.ndim == 2:
        K, y, mean = K[None], y[None], mean[None]
    n = K.shape[-1]
    if raw_noise.grad is not None:
        raw_noise.grad = None
    sigma2 = F.softplus(raw_noise) + 1e-4                       # GreaterThan(1e-4) constraint
    Ks = K + sigma2.reshape(-1, 1, 1) * torch.eye(n, dtype=K.dtype)
    L = torch.linalg.cholesky(Ks)
    r = (y - mean).unsqueeze(-1)
    z = torch.linalg.solve_triangular(L, r, upper=False)
    quad = (z * z).sum((-2, -1))
    logdet = 2.0 * torch.diagonal(L, dim1=-2, dim2=-1).log().sum(-1)
    mll = -0.5 * (quad + logdet + n * LOG_2PI) / n              # per-datum, like gpytorch
    loss = -mll.sum()
    loss.backward()
    return mll.detach(), -raw_noise.grad.detach().clone()


def mll_value_fp64(K, y, mean, raw_noise):
    """Dense fp64 Gaussian log-density via torch (used to cross-check volt_oracle by autograd)."""
    K, y, mean = K.double(), y.double(), mean.double()
    n = K.shape[-1]
    sigma2 = F.softplus(raw_noise.double()) + 1e-4
    Ks = K + sigma2 * torch.eye(n, dtype=torch.float64)
    d = torch.distributions.MultivariateNormal(mean, covariance_matrix=Ks)
    return d.log_prob(y) / n
