"""Volatility likelihood of the GPCV stage, voltron/likelihoods/volatility_likelihood.py:8-58 -- SURVEY 8(f) row 4.

``y | f ~ N(0, scale(f))`` with ``scale = clamp(exp f, 1e-3)`` ("exp", what LearnGPCV uses, train_utils.py:20) or the
copula-process warp ``sum_k a_k log(1 + exp(b_k f + c_k))`` ("cv").  ``forward`` is elementwise and stays in torch; the
expectation under q(f) that the ELBO needs (75-node Gauss-Hermite, train_utils.py:50) runs in the HIP step
(volt_gpcv_step_f32) for the "exp" parameterisation -- the "cv" one has no accelerated ELBO here."""
import torch
import torch.nn.functional as F
from torch import nn
from torch.distributions import Normal

from ..gp import Module, MultivariateNormal

NUM_LIKELIHOOD_SAMPLES = 10       # gpytorch.settings.num_likelihood_samples default, used by Likelihood.marginal


class VolatilityGaussianLikelihood(Module):
    MIN_SCALE = 1e-3              # volatility_likelihood.py:50

    def __init__(self, K=5, batch_shape=torch.Size(), param="cv", *args, **kwargs):
        super().__init__()
        if param == "cv":
            self.raw_a = nn.Parameter(torch.rand(*batch_shape, K))
            self.raw_b = nn.Parameter(0.1 * torch.rand(*batch_shape, K))
            self.raw_c = nn.Parameter(torch.rand(*batch_shape, K))
        self.param = param

    # constraints of volatility_likelihood.py:24-26: Positive (softplus), Interval(0,3), Interval(-3,3) (sigmoid)
    @property
    def trans_a(self):
        return F.softplus(self.raw_a)

    @property
    def trans_b(self):
        return 3.0 * torch.sigmoid(self.raw_b)

    @property
    def trans_c(self):
        return 6.0 * torch.sigmoid(self.raw_c) - 3.0

    def forward(self, function_samples, *args, **kwargs):
        if self.param == "cv":
            transform = ((self.trans_b * function_samples.unsqueeze(-1) + self.trans_c).exp() + 1).log() * self.trans_a
            summed_transform = transform.sum(-1)
        else:
            summed_transform = function_samples.exp()
        return Normal(torch.zeros_like(summed_transform), summed_transform.clamp(min=self.MIN_SCALE))

    def expected_log_prob(self, target, input, *params, **kwargs):
        """volatility_likelihood.py:52-57 (gpytorch ``_OneDimensionalLikelihood.expected_log_prob``): E_{q(f_i)} log p(y_i|f_i)
        per point by Gauss-Hermite quadrature with ``num_gauss_hermite_locs`` nodes.  O(N Q) elementwise, kept in
        torch for direct callers; LearnGPCV's loop gets the same numbers (and their gradients) from the fused HIP step."""
        import math
        from ..variational import _gauss_hermite, num_gauss_hermite_locs
        gx, gw = _gauss_hermite(num_gauss_hermite_locs.value(), target.device)
        mean, var = input.mean, input.variance
        locs = torch.sqrt(2.0 * var).unsqueeze(-1) * gx + mean.unsqueeze(-1)
        logp = self.forward(locs.movedim(-1, 0)).log_prob(target)               # [Q, ..., N]
        res = (logp * gw.reshape(-1, *([1] * (logp.ndim - 1)))).sum(0)
        return res

    def marginal(self, function_dist, *args, **kwargs):
        """gpytorch Likelihood.marginal: push ``num_likelihood_samples`` joint draws of f through ``forward``."""
        samples = function_dist.rsample(torch.Size([NUM_LIKELIHOOD_SAMPLES]))
        return self.forward(samples, *args, **kwargs)

    def __call__(self, input, *args, **kwargs):
        if isinstance(input, MultivariateNormal):
            return self.marginal(input, *args, **kwargs)
        return self.forward(input, *args, **kwargs)
