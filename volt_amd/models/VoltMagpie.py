"""VoltMagpie -- the ``voltron.models.VoltMagpie`` surface (voltron/models/VoltMagpie.py:15-127): moving-average mean
(EWMA, k taps) + volatility kernel.  State, caching and prediction methods live in models/_base.py."""
from ..means import EWMAMean, DEWMAMean, TEWMAMean            # noqa: F401  (importable from here, as in the reference)
from ._base import VolGP


class VoltMagpie(VolGP):
    def __init__(self, train_x, train_y, likelihood, vol_path=None, k=25):
        super().__init__(train_x, train_y, likelihood)
        self.mean_module = EWMAMean(train_x, train_y, k).to(train_x.device)
        self._init_vol_state(train_x, train_y, vol_path)
