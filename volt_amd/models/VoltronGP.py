"""VoltronGP -- drop-in for voltron/models/VoltronGP.py (LinearMean + volatility kernel); the
notebook's ``TrainDataModel`` swaps in a LogLinearMean (train_utils.py:102)."""
import torch

from ..gp import ExactGP, ExactMarginalLogLikelihood, GaussianLikelihood, LinearMean, MultivariateNormal, same_values
from ..kernels import VolatilityKernel
from .BMGP import BMGP


class VoltronGP(ExactGP):
    def __init__(self, train_x, train_y, likelihood, vol_path=None):
        # WE ASSUME IN THE BATCHED CASE THAT  TRAIN_X: N,  TRAIN_Y: T X N,  VOL_PATH: T X N
        super(VoltronGP, self).__init__(train_x, train_y, likelihood)

        if train_y.ndim > 1:
            batch_shape = train_y.shape[:-1]
        else:
            batch_shape = torch.Size()

        self.mean_module = LinearMean(1, batch_shape=batch_shape).to(train_x.device)
        self.covar_module = VolatilityKernel()

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
                                           self.log_vol_path.exp().unsqueeze(-1)).detach()     # VoltronGP.py:41

        if batch_shape == torch.Size():
            self.vol_lh = GaussianLikelihood().to(train_x.device)
            self.vol_model = BMGP(train_x, self.log_vol_path, self.vol_lh)
        else:
            self.vol_lh = GaussianLikelihood(batch_shape=batch_shape).to(train_x.device)
            self.vol_model = None          # reference: botorch MultitaskBMGP, out of scope

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
        return _model_generate_prediction(self, test_x, pred_vol, n_sample)                    # VoltronGP.py:62-95

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
