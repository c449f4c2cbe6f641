"""voltron/means/loglinear_mean.py: log(clamp(w x + b)) -- trivial elementwise, stays in torch."""
import torch

from ..gp import LinearMean


class LogLinearMean(LinearMean):
    def __init__(self, input_size, batch_shape=None, bias=True):
        if batch_shape is None:
            batch_shape = torch.Size()
        super().__init__(input_size=input_size, batch_shape=batch_shape, bias=bias)

    def initialize_from_data(self, x, y):
        with torch.no_grad():
            self.bias.data = y.exp().mean(-1, keepdim=True)       # y is on the log scale (:13-16)

    def forward(self, x):
        linear_term = super().forward(x)
        return linear_term.clamp(min=1e-6).log()                   # :19-21
