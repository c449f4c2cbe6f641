"""Moving-average means -- drop-in for voltron/means/EWMA.py with the convolution in libvolt_hip.so.

Same classes, constructor arguments and three-way ``forward`` return as the reference (:46-54):
one-point query -> last value, query equal to train_x -> ma[:-1], anything else -> the whole
N+1 vector.  The reference forces every result through ``.type(torch.FloatTensor)`` (a host
round trip, :37,:50-54) and then back ``.to(self.train_x.device)``; here results are fp32 and
stay on ``train_x``'s device.
"""
import numpy as np
import torch

from .. import ops
from ..gp import Mean, same_values


def EWMA(y, k):
    """EWMA.py:20-37: [..., N] -> [..., N+1], out[t] = sum_j w_j padded[t+j], padded = k x y[0] ++ y."""
    return ops.ewma(y, k)


class _MAMean(Mean):
    def __init__(self, train_x, train_y, k=20):
        super().__init__()
        self.k = k
        self.train_x = train_x
        self.train_y = train_y

    def _select(self, ma, x):
        dev = self.train_x.device
        if x.numel() == 1:
            return ma[..., -1].unsqueeze(0).to(dev)
        elif same_values(x.squeeze(), self.train_x.squeeze()):
            return ma[..., :-1].to(dev)
        else:
            return ma.to(dev)


class EWMAMean(_MAMean):
    def forward(self, x):                               # EWMA.py:46-54
        return self._select(EWMA(self.train_y, self.k), x)


class HEWMAMean(_MAMean):
    def forward(self, x):                               # EWMA.py:64-71 (unused by the reference's drivers)
        wma_k = EWMA(self.train_y, self.k)
        wma_k2 = EWMA(self.train_y, int(self.k / 2))
        hma = EWMA(2 * wma_k2[:-1] - wma_k[:-1], int(np.sqrt(self.k)))
        if same_values(x.squeeze(), self.train_x.squeeze()):
            return hma[:-1].to(self.train_x.device)
        return hma.to(self.train_x.device)


class DEWMAMean(_MAMean):
    def forward(self, x):                               # EWMA.py:81-91
        ema = EWMA(self.train_y, self.k)
        ema_ema = EWMA(ema, self.k)[..., :-1]
        return self._select(2 * ema - ema_ema, x)


class TEWMAMean(_MAMean):
    def __init__(self, train_x, train_y, k=20):
        super().__init__(train_x, train_y, k)
        self.alpha = 2. / (self.k + 1)

    def forward(self, x):                               # EWMA.py:102-113
        ema = EWMA(self.train_y, self.k)
        ema_ema = EWMA(ema, self.k)[..., :-1]
        ema_ema_ema = EWMA(ema_ema, self.k)[..., :-1]
        return self._select(3 * ema - 3 * ema_ema + ema_ema_ema, x)


class MeanRevertingEMAMean(_MAMean):
    def __init__(self, train_x, train_y, k=20, theta=0.5):
        super().__init__(train_x, train_y, k)
        self.theta = theta
        self.alpha = 2. / (self.k + 1)
        self.latent_mean = train_y.mean()               # fixed at construction (EWMA.py:124)

    def forward(self, x):                               # EWMA.py:126-135
        ema = EWMA(self.train_y, self.k)
        ema[..., 1:] -= self.theta * (ema[..., :-1] - self.latent_mean)
        return self._select(ema, x)
