"""Ornstein-Uhlenbeck kernel, voltron/kernels/OUKernel.py:8-33 (exported by voltron.kernels, used by none of the
reference's drivers): k = exp(-|x1 - x2| / (2 l)).  Elementwise torch with autograd.  Quirk NOT mirrored: the reference's
fast branch routes the *unsquared* distance through gpytorch's ``RBFCovariance`` function, whose hand-written backward
assumes a squared distance, so its lengthscale gradient is that of a different function; the values are the same."""
import torch

from ..gpkernels import _Stationary


class OUKernel(_Stationary):
    has_lengthscale = True

    def forward(self, x1, x2=None, diag=False, **params):
        k = torch.exp(-0.5 * self._scaled_dist(x1, x2))
        return k.diagonal(dim1=-2, dim2=-1) if diag else k
