"""Fractional-Brownian-motion kernel, voltron/kernels/FBMKernel.py:6-58 (LearnGPCV's ``kernel="fbm"`` option,
train_utils.py:24-25).  cov = (|x1|^2H + |x2|^2H - |x1-x2|^2H)/2 with H = ``vol`` in (0,1) through a sigmoid.
Elementwise and differentiable in torch; the dense d ELBO / d K it is contracted with comes from the HIP step."""
import torch
from torch import nn

from ..gp import Kernel


class FBMKernel(Kernel):
    has_lengthscale = False

    def __init__(self, vol=0.2, batch_shape=None, vol_constraint=None, **kwargs):
        super().__init__(**kwargs)
        if batch_shape is None:
            batch_shape = torch.Size()
            vol_size = [1]
        else:
            vol_size = [*batch_shape, 1]
        self.batch_shape = batch_shape
        self.register_parameter("raw_vol", nn.Parameter(torch.zeros(*vol_size)))
        self.vol = vol

    @property
    def vol(self):
        return torch.sigmoid(self.raw_vol)                      # Interval(0., 1.).transform

    @vol.setter
    def vol(self, value):
        value = torch.as_tensor(value, dtype=self.raw_vol.dtype, device=self.raw_vol.device)
        with torch.no_grad():
            self.raw_vol.copy_(torch.logit(value).expand_as(self.raw_vol))

    def forward(self, x1s, x2s=None, **kwargs):
        if x2s is None:
            x2s = x1s
        x1s = x1s.unsqueeze(1)                                  # FBMKernel.py:39-40
        x2s = x2s.unsqueeze(0)
        double_vol = 2. * self.vol
        if self.batch_shape != torch.Size():                    # T exponents over shared inputs -> [T,N,N]
            double_vol = double_vol.reshape(-1, 1, 1, 1)
        dist = x1s.abs().pow(double_vol) + x2s.abs().pow(double_vol) - (x1s - x2s).abs().pow(double_vol)
        cov = dist.squeeze(-1) / 2.
        if kwargs.pop("diag", False):
            return cov.diag()
        return cov
