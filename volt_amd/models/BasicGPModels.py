"""Baseline exact GPs, voltron/models/BasicGPModels.py:7-28 -- SURVEY 8(f) row 2 (the models ``nonvol_rollouts``
and ``TrainBasicModel`` work on).  Same constructor arguments and module names; the MLL, its gradient (dense
d mll / d K through the kernel's elementwise autograd), the eval-mode posterior and ``posterior(X)`` run on the HIP
library (gp.exact_posterior).  ``posterior`` is the slice of botorch's ``GPyTorchModel`` interface that
rollout_utils.py:99,114 uses (latent f, ``observation_noise=False``)."""
import torch

from ..gp import ConstantMean, ExactGP, GPPosterior, MultivariateNormal, exact_posterior
from ..gpkernels import MaternKernel, ScaleKernel, SpectralMixtureKernel


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
