"""Stationary kernels of the baseline GPs (voltron/models/BasicGPModels.py:10,20; experiments/weather/BasicWind.py:30-38)
-- outside the product package (out of scope), kept for the (f)2 tests.  These are gpytorch classes (ScaleKernel, RBFKernel, MaternKernel, SpectralMixtureKernel),
restated from their published formulas with gpytorch's parameter names and ``Positive`` (softplus) constraints; the
covariance entries are elementwise torch (autograd gives dK/dtheta), the factorisation, the MLL and the dense
d mll / d K they are contracted with run in libvolt_hip.so (volt_mll_step_f32 + volt_mll_grad_k_f32)."""
import math

import torch
import torch.nn.functional as F
from torch import nn

from volt_amd.gp import Kernel, _dense


def _inv_softplus(v):
    return v + torch.log(-torch.expm1(-v))


def _col(x):
    return x.unsqueeze(-1) if x.ndim == 1 else x


class _Stationary(Kernel):
    has_lengthscale = True

    def __init__(self, ard_num_dims=None, batch_shape=torch.Size(), **kwargs):
        super().__init__(**kwargs)
        d = 1 if ard_num_dims is None else ard_num_dims
        self.batch_shape = batch_shape
        self.register_parameter("raw_lengthscale", nn.Parameter(torch.zeros(*batch_shape, 1, d)))

    @property
    def lengthscale(self):
        return F.softplus(self.raw_lengthscale)

    @lengthscale.setter
    def lengthscale(self, value):
        value = torch.as_tensor(value, dtype=self.raw_lengthscale.dtype, device=self.raw_lengthscale.device)
        with torch.no_grad():
            self.raw_lengthscale.copy_(_inv_softplus(value).expand_as(self.raw_lengthscale))

    def _scaled_dist(self, x1, x2, squared=False):
        x1, x2 = _col(x1), _col(x1 if x2 is None else x2)
        a, b = x1 / self.lengthscale, x2 / self.lengthscale
        d2 = (a.unsqueeze(-2) - b.unsqueeze(-3)).pow(2).sum(-1)
        return d2 if squared else d2.clamp_min(1e-30).sqrt()


class RBFKernel(_Stationary):
    def forward(self, x1, x2=None, diag=False, **params):
        k = torch.exp(-0.5 * self._scaled_dist(x1, x2, squared=True))
        return k.diagonal(dim1=-2, dim2=-1) if diag else k


class MaternKernel(_Stationary):
    def __init__(self, nu=2.5, **kwargs):
        if nu not in {0.5, 1.5, 2.5}:
            raise RuntimeError("nu expected to be 0.5, 1.5, or 2.5")
        super().__init__(**kwargs)
        self.nu = nu

    def forward(self, x1, x2=None, diag=False, **params):
        d = self._scaled_dist(x1, x2)
        e = torch.exp(-math.sqrt(self.nu * 2) * d)
        if self.nu == 0.5:
            c = 1.0
        elif self.nu == 1.5:
            c = 1.0 + math.sqrt(3) * d
        else:
            c = 1.0 + math.sqrt(5) * d + 5.0 / 3.0 * d ** 2
        k = c * e
        return k.diagonal(dim1=-2, dim2=-1) if diag else k


class ScaleKernel(Kernel):
    def __init__(self, base_kernel, batch_shape=torch.Size(), **kwargs):
        super().__init__(**kwargs)
        self.base_kernel = base_kernel
        self.register_parameter("raw_outputscale", nn.Parameter(torch.zeros(torch.Size(batch_shape))))

    @property
    def outputscale(self):
        return F.softplus(self.raw_outputscale)

    @outputscale.setter
    def outputscale(self, value):
        value = torch.as_tensor(value, dtype=self.raw_outputscale.dtype, device=self.raw_outputscale.device)
        with torch.no_grad():
            self.raw_outputscale.copy_(_inv_softplus(value).expand_as(self.raw_outputscale))

    def forward(self, x1, x2=None, diag=False, **params):
        k = _dense(self.base_kernel(x1, x2, diag=diag, **params))
        o = self.outputscale
        return k * (o.reshape(*o.shape, 1) if diag else o.reshape(*o.shape, 1, 1))


class SpectralMixtureKernel(Kernel):
    """k(tau) = sum_q w_q prod_d exp(-2 pi^2 tau_d^2 v_qd^2) cos(2 pi tau_d mu_qd)."""

    def __init__(self, num_mixtures=None, ard_num_dims=1, batch_shape=torch.Size(), **kwargs):
        if num_mixtures is None:
            raise RuntimeError("num_mixtures is a required argument")
        super().__init__(**kwargs)
        self.num_mixtures, self.ard_num_dims = num_mixtures, ard_num_dims
        self.register_parameter("raw_mixture_weights", nn.Parameter(torch.zeros(*batch_shape, num_mixtures)))
        ms = torch.Size([*batch_shape, num_mixtures, 1, ard_num_dims])
        self.register_parameter("raw_mixture_means", nn.Parameter(torch.zeros(ms)))
        self.register_parameter("raw_mixture_scales", nn.Parameter(torch.zeros(ms)))

    mixture_weights = property(lambda self: F.softplus(self.raw_mixture_weights))
    mixture_means = property(lambda self: F.softplus(self.raw_mixture_means))
    mixture_scales = property(lambda self: F.softplus(self.raw_mixture_scales))

    def _set(self, raw, value):
        value = torch.as_tensor(value, dtype=raw.dtype, device=raw.device)
        with torch.no_grad():
            raw.copy_(_inv_softplus(value).expand_as(raw))

    def initialize_from_data(self, train_x, train_y, **kwargs):
        """gpytorch's heuristic: scales ~ 1/|N(0, max_dist^2)|, means ~ U(0, 0.5/min_dist), weights = std(y)/Q."""
        with torch.no_grad():
            train_x = _col(train_x)
            xs = train_x.sort(dim=-2)[0]
            max_dist = xs[..., -1, :] - xs[..., 0, :]
            dists = xs[..., 1:, :] - xs[..., :-1, :]
            dists = torch.where(dists.eq(0.0), torch.tensor(1.0e10, dtype=train_x.dtype, device=train_x.device), dists)
            min_dist = dists.sort(dim=-2)[0][..., 0, :]
            pd = self.raw_mixture_scales.device               # parameters may still be on the host at construction
            max_dist, min_dist, train_y = max_dist.to(pd), min_dist.to(pd), train_y.to(pd)
            self._set(self.raw_mixture_scales,
                      torch.randn_like(self.raw_mixture_scales).mul_(max_dist).abs_().reciprocal_())
            self._set(self.raw_mixture_means, torch.rand_like(self.raw_mixture_means).mul_(0.5).div(min_dist))
            self._set(self.raw_mixture_weights, train_y.std().div(self.num_mixtures))

    def forward(self, x1, x2=None, diag=False, **params):
        x1, x2 = _col(x1), _col(x1 if x2 is None else x2)
        a, b = x1.unsqueeze(-3), x2.unsqueeze(-3)                                  # [1,n,d]
        tau = a.unsqueeze(-2) - b.unsqueeze(-3)                                    # [1,n,m,d]
        sc, mu = self.mixture_scales.unsqueeze(-2), self.mixture_means.unsqueeze(-2)   # [Q,1,1,d]
        res = (torch.exp(-2 * math.pi ** 2 * (tau * sc) ** 2) * torch.cos(2 * math.pi * tau * mu)).prod(-1)   # [Q,n,m]
        w = self.mixture_weights
        k = (res * w.reshape(*w.shape, 1, 1)).sum(-3)
        return k.diagonal(dim1=-2, dim2=-1) if diag else k
