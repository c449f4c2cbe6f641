"""voltron/means/mulidentity_mean.py:6-20: mean(x) = constant * x (elementwise; exported, unused by the drivers)."""
import torch
from torch import nn

from ..gp import Mean


class MulIdentityMean(Mean):
    def __init__(self, prior=None, batch_shape=torch.Size(), **kwargs):
        super().__init__()
        self.batch_shape = batch_shape
        self.register_parameter(name="constant", param=nn.Parameter(torch.zeros(*batch_shape, 1)))
        if prior is not None:
            self.register_prior("mean_prior", prior, "constant")

    def forward(self, input):
        return (self.constant * input).squeeze(-1)
