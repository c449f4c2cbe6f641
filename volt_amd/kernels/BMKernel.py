"""Brownian-motion kernel, voltron/kernels/BMKernel.py:6-52 -- SURVEY 8(f) row 1 ("next": the
vol-path forecaster).  cov = vol * min(x1, x2); `vol` constrained to (0,1) through a sigmoid like
gpytorch's Interval(0., 1.).  Elementwise, kept in torch; the factorisations it feeds run in HIP."""
import torch
from torch import nn

from ..gp import Kernel, _ScaledDense


class BMKernel(Kernel):
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
        return torch.sigmoid(self.raw_vol)                      # Interval(0,1).transform

    @vol.setter
    def vol(self, value):
        value = torch.as_tensor(value, dtype=self.raw_vol.dtype, device=self.raw_vol.device)
        with torch.no_grad():
            self.raw_vol.copy_(torch.logit(value).expand_as(self.raw_vol))

    def forward(self, x1s, x2s=None, **kwargs):
        if x2s is None:
            x2s = x1s
        if x1s.ndim == 1:
            x1s = x1s.unsqueeze(-1)
        if x2s.ndim == 1:
            x2s = x2s.unsqueeze(-1)
        if self.batch_shape == torch.Size():
            cov = self.vol * torch.minimum(x1s[:, 0].unsqueeze(-1), x2s[:, 0].unsqueeze(-2))
        else:
            m = torch.minimum(x1s[0, :, 0].unsqueeze(-1), x2s[0, :, 0].unsqueeze(-2))
            cov = self.vol.unsqueeze(-1) * m.unsqueeze(0).repeat(*self.batch_shape, 1, 1)
        if kwargs.pop("diag", False):
            return cov.diag()
        return cov

    def __call__(self, x1, x2=None, **kwargs):
        """Lazy form: keeps K = vol * min(x1, x2) factored so an exact MLL can differentiate wrt vol."""
        x2 = x1 if x2 is None else x2
        if kwargs.get("diag", False) or (self.batch_shape != torch.Size() and (x1.ndim > 2 or x2.ndim > 2)):
            return super().__call__(x1, x2, **kwargs)
        a = x1[:, 0] if x1.ndim > 1 else x1
        b = x2[:, 0] if x2.ndim > 1 else x2
        return _ScaledDense(self.vol, torch.minimum(a.unsqueeze(-1), b.unsqueeze(-2)))
