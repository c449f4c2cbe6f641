"""VoltronGP -- the ``voltron.models.VoltronGP`` surface (voltron/models/VoltronGP.py:11-123): linear mean +
volatility kernel (the notebook's TrainDataModel swaps in a LogLinearMean, train_utils.py:102)."""
from ..gp import LinearMean
from ._base import VolGP


class VoltronGP(VolGP):
    def __init__(self, train_x, train_y, likelihood, vol_path=None):
        super().__init__(train_x, train_y, likelihood)
        self.mean_module = LinearMean(1, batch_shape=train_y.shape[:-1]).to(train_x.device)
        self._init_vol_state(train_x, train_y, vol_path)
