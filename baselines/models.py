"""Baseline exact GPs, voltron/models/BasicGPModels.py:7-28 -- OUTSIDE the product package (SURVEY 2 row 13 marks them
out of scope): they are the generic-kernel models the (f)2 tests hand to ``nonvol_rollouts``.  Same constructor arguments and module names; the MLL, its gradient (dense
d mll / d K through the kernel's elementwise autograd), the eval-mode posterior and ``posterior(X)`` run on the HIP
library (gp.exact_posterior).  ``posterior`` is the slice of botorch's ``GPyTorchModel`` interface that
rollout_utils.py:99,114 uses (latent f, ``observation_noise=False``)."""
import torch

from volt_amd import ops
from volt_amd.gp import ConstantMean, ExactGP, MultivariateNormal, _dense, _safe_factor

from .gpkernels import MaternKernel, ScaleKernel, SpectralMixtureKernel


def exact_posterior(model, x, observation_noise=False):
    """Exact-GP predictive at x for a model with ``mean_module(x)`` and a two-input ``covar_module(x1, x2)``:
    mean = m(x) + K_*t K_s^-1 (y - m(X)),  cov = K_** - K_*t K_s^-1 K_t*  (+ noise).  K_s = L L' on the HIP potrf,
    K_s^-1 = Y Y' with Y = L^-T from the HIP triangular inverse, products on the library GEMM.
    Targets [S,N] (one shared input set, S target vectors -- nonvol_rollouts' stacked samples) give a batch mean."""
    with torch.no_grad():
        xt = model.train_inputs[0]
        x = x.unsqueeze(-1) if x.ndim == 1 else x
        y = model.train_targets
        n = xt.shape[-2]
        Ktt = _dense(model.covar_module(xt, xt)).to(torch.float32)
        noise = model.likelihood.noise.reshape(-1)[:1]
        A = (Ktt + noise * torch.eye(n, device=xt.device)).reshape(1, n, n)
        f, _ = _safe_factor(A)
        Y = ops.trtri(f)[0]                                           # L^-T  (upper)
        Kst = _dense(model.covar_module(x, xt)).to(torch.float32)     # [H,N]
        G = ops.gemm_nt(Kst, Y.mT.contiguous(), uplo_b=1)             # K_*t L^-T
        r = (y - model.mean_module(xt)).to(torch.float32).reshape(-1, n)
        z = ops.gemm_nt(r, Y.mT.contiguous(), uplo_b=1)               # rows L^-1 r
        mean = model.mean_module(x) + ops.gemm_nt(z, G).reshape(*y.shape[:-1], x.shape[-2])
        cov = _dense(model.covar_module(x, x)).to(torch.float32) - ops.gemm_nt(G, G)
        if observation_noise:
            cov = cov + noise * torch.eye(x.shape[-2], device=x.device)
        return MultivariateNormal(mean, cov)


class GPPosterior:
    """What botorch's ``model.posterior(X)`` hands back, as far as rollout_utils.py:99,114 uses it:
    ``.mean`` / ``.variance`` [.., q, 1] and ``.sample(sample_shape)`` -> sample_shape x .. x q x 1."""

    def __init__(self, mvn):
        self.mvn = mvn

    @property
    def mean(self):
        return self.mvn.mean.unsqueeze(-1)

    @property
    def variance(self):
        return self.mvn.variance.unsqueeze(-1)

    def rsample(self, sample_shape=torch.Size(), base_samples=None):
        return self.mvn.rsample(sample_shape, base_samples).unsqueeze(-1)

    def sample(self, sample_shape=torch.Size(), base_samples=None):
        with torch.no_grad():
            return self.rsample(sample_shape, base_samples)


class _DenseKernelGP(ExactGP):
    def forward(self, x):
        mean_x = self.mean_module(x)
        covar_x = self.covar_module(x)
        return MultivariateNormal(mean_x, covar_x)

    def posterior_call(self, x):
        return exact_posterior(self, x)

    def posterior(self, X, observation_noise=False, **kwargs):
        return GPPosterior(exact_posterior(self, X, observation_noise=observation_noise))


class MaternGP(_DenseKernelGP):
    def __init__(self, train_x, train_y, likelihood):
        super(MaternGP, self).__init__(train_x, train_y, likelihood)
        self.mean_module = ConstantMean()
        self.covar_module = ScaleKernel(MaternKernel())


class SMGP(_DenseKernelGP):
    def __init__(self, train_x, train_y, likelihood, num_mixtures=10):
        super(SMGP, self).__init__(train_x, train_y, likelihood)
        self.mean_module = ConstantMean()
        self.covar_module = SpectralMixtureKernel(num_mixtures=num_mixtures)
        self.covar_module.initialize_from_data(train_x, train_y)
