"""VoltMagpie -- drop-in for voltron/models/VoltMagpie.py (EWMA mean + volatility kernel).

Same constructor, attributes (train_x, train_y, log_vol_path, train_cov, mean_module,
covar_module, vol_lh, vol_model, likelihood) and methods.  ``train_cov`` is filled ONCE on the
device by the HIP kernels (VoltMagpie.py:46) and reused by every training step (:123-124).
Batched layout as documented there: train_x [N], train_y [T,N], vol_path [T,N].
"""
import torch

from ..gp import ExactGP, ExactMarginalLogLikelihood, GaussianLikelihood, MultivariateNormal, same_values
from ..kernels import VolatilityKernel
from ..means import EWMAMean, DEWMAMean, TEWMAMean            # noqa: F401  (re-exported like the reference)
from .BMGP import BMGP


class VoltMagpie(ExactGP):
    def __init__(self, train_x, train_y, likelihood, vol_path=None, k=25):
        # WE ASSUME IN THE BATCHED CASE THAT  TRAIN_X: N,  TRAIN_Y: T X N,  VOL_PATH: T X N
        super(VoltMagpie, self).__init__(train_x, train_y, likelihood)

        if train_y.ndim > 1:
            batch_shape = train_y.shape[:-1]
        else:
            batch_shape = torch.Size()

        self.mean_module = EWMAMean(train_x, train_y, k).to(train_x.device)
        self.covar_module = VolatilityKernel().to(train_x.device)

        if train_y.ndim > 1:
            self.train_x = train_x.unsqueeze(0).repeat(*batch_shape, 1)
        else:
            self.train_x = train_x
        self.train_y = train_y

        if vol_path is None:
            self.log_vol_path = -1 * torch.ones(train_x.shape[0], device=train_x.device)
        else:
            self.log_vol_path = vol_path.log()

        self.train_cov = self.covar_module(self.train_x.unsqueeze(-1),
                                           self.log_vol_path.exp().unsqueeze(-1)).detach()

        if batch_shape == torch.Size():
            self.vol_lh = GaussianLikelihood().to(train_x.device)
            self.vol_model = BMGP(train_x, self.log_vol_path, self.vol_lh)
        else:
            # reference: botorch KroneckerMultiTaskGP (MultitaskBMGP) -- out of scope; the caller
            # supplies vol_model/vol_lh (train_utils.py:223-224) when it needs vol forecasts.
            self.vol_lh = GaussianLikelihood(batch_shape=batch_shape).to(train_x.device)
            self.vol_model = None

    def UpdateVolPath(self, vol_path):
        self.log_vol_path = vol_path.log()
        self.train_cov = self.covar_module(self.train_x, self.log_vol_path.exp())
        return

    def VolMLL(self):
        vol_mll = ExactMarginalLogLikelihood(self.vol_lh, self.vol_model)
        outputs = self.vol_model(self.train_x)
        return vol_mll(outputs, self.log_vol_path)

    def GeneratePrediction(self, test_x, pred_vol, n_sample=1):
        from ..rollout_utils import _model_generate_prediction
        return _model_generate_prediction(self, test_x, pred_vol, n_sample)      # VoltMagpie.py:67-99

    def SamplePrediction(self, test_x, n_sample=1, return_vol=False):
        self.vol_model.eval()
        pred_vol = self.vol_model(test_x).sample().exp()
        if pred_vol.ndim > 1:                    # reference: .transpose(-1, -2) unconditionally, which only
            pred_vol = pred_vol.transpose(-1, -2)   # works for its multitask vol model ([H, T] samples)
        prediction = self.GeneratePrediction(test_x, pred_vol, n_sample)
        if return_vol:
            return prediction, pred_vol
        return prediction

    def MeanPrediction(self, test_x, n_sample=1, return_vol=False):
        self.vol_model.eval()
        pred_vol = self.vol_model(test_x).mean.exp()
        if pred_vol.ndim > 1:
            pred_vol = pred_vol.transpose(-1, -2)
        prediction = self.GeneratePrediction(test_x, pred_vol, n_sample)
        if return_vol:
            return prediction, pred_vol
        return prediction

    def forward(self, x):
        mean_x = self.mean_module(x)
        if same_values(x, self.train_inputs[0]):                  # torch.equal without the device sync when aliased
            covar_x = self.train_cov
        else:
            covar_x = self.covar_module(x, self.log_vol_path.exp())
        return MultivariateNormal(mean_x, covar_x)
