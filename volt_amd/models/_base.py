"""What VoltronGP, VoltMagpie and Volt share (voltron/models/{VoltronGP,VoltMagpie,Volt}.py): an exact GP whose
covariance is the volatility kernel of a stored vol path, filled ONCE on the device (``train_cov``) and reused by every
training step, plus a Brownian-motion GP over the log-vol path that forecasts it.  Attribute names are the reference's
(rollout_utils.py:7-32,81-86 and train_utils.py:222-224 read and write them); the subclasses only choose the mean."""
import torch

from ..gp import ExactGP, ExactMarginalLogLikelihood, GaussianLikelihood, MultivariateNormal, same_values
from ..kernels import VolatilityKernel
from .BMGP import BMGP


class VolGP(ExactGP):
    def _init_vol_state(self, x, y, vol_path):
        """x [N], y [N] or [T,N] (batched layout: one shared input grid, T target / vol rows), vol_path like y or None.
        Call after ``mean_module`` is set so that the modules register in the reference's order."""
        dev = x.device
        batch_shape = y.shape[:-1]
        self.covar_module = VolatilityKernel().to(dev)
        self.train_x = x.unsqueeze(0).repeat(*batch_shape, 1) if len(batch_shape) else x
        self.train_y = y
        self.log_vol_path = vol_path.log() if vol_path is not None else -torch.ones(x.shape[0], device=dev)
        self.train_cov = self.covar_module(self.train_x.unsqueeze(-1), self.log_vol_path.exp().unsqueeze(-1)).detach()
        # vol forecaster: one BM-GP per series.  (The reference's batched models use botorch's Kronecker multitask GP
        # there, out of scope; the batched BMGP -- T independent vol models over the shared grid -- takes its place,
        # so SamplePrediction / MeanPrediction work on batched models too.)
        self.vol_lh = GaussianLikelihood(batch_shape=batch_shape).to(dev)
        self.vol_model = BMGP(x, self.log_vol_path, self.vol_lh) if self.log_vol_path.shape[:-1] == batch_shape else None

    def UpdateVolPath(self, vol_path):
        self.log_vol_path = vol_path.log()
        self.train_cov = self.covar_module(self.train_inputs[0].squeeze(-1) if self.train_x.ndim == 1 else self.train_x,
                                           self.log_vol_path.exp())

    def VolMLL(self):
        vol_mll = ExactMarginalLogLikelihood(self.vol_lh, self.vol_model)
        return vol_mll(self.vol_model(self.train_x), self.log_vol_path)

    def GeneratePrediction(self, test_x, pred_vol, n_sample=1):
        from ..rollout_utils import _model_generate_prediction
        return _model_generate_prediction(self, test_x, pred_vol, n_sample)

    def _predict_with(self, pred_vol, test_x, n_sample, return_vol):
        prediction = self.GeneratePrediction(test_x, pred_vol, n_sample)
        return (prediction, pred_vol) if return_vol else prediction

    def SamplePrediction(self, test_x, n_sample=1, return_vol=False):
        self.vol_model.eval()
        return self._predict_with(self.vol_model(test_x).sample().exp(), test_x, n_sample, return_vol)

    def MeanPrediction(self, test_x, n_sample=1, return_vol=False):
        self.vol_model.eval()
        return self._predict_with(self.vol_model(test_x).mean.exp(), test_x, n_sample, return_vol)

    def forward(self, x):
        mean_x = self.mean_module(x)
        if same_values(x, self.train_inputs[0]):          # torch.equal without the device sync when aliased
            covar_x = self.train_cov                      # the cached fill
        else:
            covar_x = self.covar_module(x, self.log_vol_path.exp())
        return MultivariateNormal(mean_x, covar_x)
