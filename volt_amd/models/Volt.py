"""Volt -- drop-in for voltron/models/Volt.py (all-in-one Train / Forecast class; the reference does
not export it from models/__init__.py either).

``Train`` runs the reference's three stages (Volt.py:95-146) on the HIP path: LearnGPCV (variational
volatility extraction), TrainVolModel (BM-GP over log-vol) and the exact-GP data model; ``Train(vol=...)``
skips the first two when a volatility path is supplied.

Deliberate deviations, each a reference defect (SURVEY 7 hard part 5):
* ``Forecast`` in the reference passes ``return_vol`` / ``latent_mean`` keywords that ``Rollouts``
  does not accept (Volt.py:155-160 vs rollout_utils.py:57) and would raise TypeError; here it calls
  ``Rollouts`` with the arguments it does accept (``theta`` only when mean_revert is set).
"""
import torch

from ..gp import ConstantMean, GaussianLikelihood
from ..means import EWMAMean, DEWMAMean, TEWMAMean
from ..rollout_utils import Rollouts
from ._base import VolGP

_MEAN_CLASSES = {"ewma": EWMAMean, "dewma": DEWMAMean, "tewma": TEWMAMean}


class Volt(VolGP):
    def __init__(self, train_x, log_data, mean='constant', vol_path=None, k=25):
        # The reference builds the ExactGP on [1:] but keeps the FULL train_x / log_data as attributes (Volt.py:52-62);
        # with a vol_path of length N-1 its own train_cov line is then shape-inconsistent.  The [1:] view is kept
        # everywhere here, so that train_cov matches train_inputs.
        x, y = train_x[1:], log_data[..., 1:]
        super().__init__(x, log_data[1:] if log_data.ndim == 1 else y, GaussianLikelihood().to(train_x.device))
        name = mean.lower()
        if name == 'constant':
            self.mean_module = ConstantMean().to(train_x.device)
        elif name in _MEAN_CLASSES:
            self.mean_module = _MEAN_CLASSES[name](x, y, k).to(train_x.device)
        else:
            raise ValueError("ERROR: Mean not implemented")      # the reference prints this and fails on the next line
        self._init_vol_state(x, y, vol_path)
        self._full_x, self._full_log_data = train_x, log_data    # Train's GPCV stage starts from the prices

    def Train(self, gpcv_iters=400, vol_mod_iters=1000, data_mod_iters=400, display=False, vol=None,
              vol_model=None, vol_lh=None):
        """Volt.py:95-146: GPCV + vol model (:103-104), then the data model (:108-146).  ``vol`` (and optionally a
        trained ``vol_model`` / ``vol_lh``) skips the first stage."""
        from ..train_utils import LR_DATA, LearnGPCV, TrainVolModel, _attach_vol, _fit_exact, _train_noise_and_mean
        x = self.train_x.squeeze()
        if vol is None:
            if self._full_log_data.ndim > 1:
                raise NotImplementedError("batched Volt.Train: fit the vol path with LearnGPCV(train_x, prices [T,N+1]) "
                                          "and pass it as vol= (the reference's MultitaskBMGP stage is out of scope)")
            vol = LearnGPCV(self._full_x[1:], self._full_log_data.exp(), gpcv_iters, printing=display)
            vol_model, vol_lh = TrainVolModel(self._full_x[1:], vol, vol_mod_iters, printing=display)
        self.UpdateVolPath(vol)
        _attach_vol(self, vol_model, vol_lh, x.device)
        params = _train_noise_and_mean(self, self.likelihood)    # noise (+ the constant mean), Volt.py:110-127
        _fit_exact(self, self.likelihood, x, self.train_y, params, LR_DATA, data_mod_iters, display)

    def Forecast(self, test_x, nsample=50, return_vol=False, mean_revert=False, theta=0.05, **rollout_kw):
        if self.vol_model is not None:
            self.vol_model.eval()
        self.eval()
        prices = torch.cat((self.train_targets[..., :1], self.train_targets.squeeze())).exp()   # Rollouts drops [0]
        return Rollouts(self.train_inputs[0].squeeze(), prices, test_x, self, nsample=nsample,
                        theta=theta if mean_revert else None, **rollout_kw)
