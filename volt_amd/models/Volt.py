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

from ..gp import ConstantMean, ExactGP, ExactMarginalLogLikelihood, GaussianLikelihood, MultivariateNormal, same_values
from ..kernels import VolatilityKernel
from ..means import EWMAMean, DEWMAMean, TEWMAMean
from ..rollout_utils import Rollouts
from .BMGP import BMGP


class Volt(ExactGP):
    def __init__(self, train_x, log_data, mean='constant', vol_path=None, k=25):
        likelihood = GaussianLikelihood().to(train_x.device)
        super(Volt, self).__init__(train_x[1:], log_data[1:], likelihood)

        if log_data.ndim > 1:
            batch_shape = log_data.shape[:-1]
        else:
            batch_shape = torch.Size()

        if mean.lower() == 'constant':
            mean_module = ConstantMean().to(train_x.device)
        elif mean.lower() == 'ewma':
            mean_module = EWMAMean(train_x[1:], log_data[1:], k).to(train_x.device)
        elif mean.lower() == 'dewma':
            mean_module = DEWMAMean(train_x[1:], log_data[1:], k).to(train_x.device)
        elif mean.lower() == 'tewma':
            mean_module = TEWMAMean(train_x[1:], log_data[1:], k).to(train_x.device)
        else:
            raise ValueError("ERROR: Mean not implemented")      # reference prints and then fails on the next line

        self.mean_module = mean_module.to(train_x.device)
        self.covar_module = VolatilityKernel().to(train_x.device)

        # NOTE (Volt.py:52-62): the reference keeps the FULL train_x / log_data here but builds the
        # ExactGP on [1:]; with a vol_path of length N-1 its own train_cov line is shape-inconsistent.
        # We keep the [1:] view everywhere so that train_cov matches train_inputs.
        if log_data.ndim > 1:
            self.train_x = train_x[1:].unsqueeze(0).repeat(*batch_shape, 1)
        else:
            self.train_x = train_x[1:]
        self.train_y = log_data[..., 1:]
        self._full_x, self._full_log_data = train_x, log_data            # Train's GPCV stage starts from the prices

        if vol_path is None:
            self.log_vol_path = -1 * torch.ones(train_x.shape[0] - 1, device=train_x.device)
        else:
            self.log_vol_path = vol_path.log()

        self.train_cov = self.covar_module(self.train_x.unsqueeze(-1),
                                           self.log_vol_path.exp().unsqueeze(-1)).detach()

        if batch_shape == torch.Size():
            self.vol_lh = GaussianLikelihood().to(train_x.device)
            self.vol_model = BMGP(train_x[1:], self.log_vol_path, self.vol_lh)
        else:
            self.vol_lh = GaussianLikelihood(batch_shape=batch_shape).to(train_x.device)
            self.vol_model = None

    def UpdateVolPath(self, vol_path):
        self.log_vol_path = vol_path.log()
        self.train_cov = self.covar_module(self.train_inputs[0], self.log_vol_path.exp())
        return

    def VolMLL(self):
        vol_mll = ExactMarginalLogLikelihood(self.vol_lh, self.vol_model)
        outputs = self.vol_model(self.train_x)
        return vol_mll(outputs, self.log_vol_path)

    def forward(self, x):
        mean_x = self.mean_module(x)
        if same_values(x, self.train_inputs[0]):                  # torch.equal without the device sync when aliased
            covar_x = self.train_cov
        else:
            covar_x = self.covar_module(x, self.log_vol_path.exp())
        return MultivariateNormal(mean_x, covar_x)

    def Train(self, gpcv_iters=400, vol_mod_iters=1000, data_mod_iters=400, display=False, vol=None,
              vol_model=None, vol_lh=None):
        """Volt.py:95-146: GPCV + vol model (:103-104), then the data model (:108-146).  ``vol`` (and optionally a
        trained ``vol_model``/``vol_lh``) skips the first stage."""
        from ..train_utils import LearnGPCV, TrainVolModel
        x = self.train_x.squeeze()
        if vol is None:
            if self._full_log_data.ndim > 1:
                raise NotImplementedError("batched Volt.Train: fit the vol path with LearnGPCV(train_x, prices [T,N+1]) "
                                          "and pass it as vol= (the reference's MultitaskBMGP stage is out of scope)")
            vol = LearnGPCV(self._full_x[1:], self._full_log_data.exp(), gpcv_iters, printing=display)
            vol_model, vol_lh = TrainVolModel(self._full_x[1:], vol, vol_mod_iters, printing=display)
        self.UpdateVolPath(vol)
        if isinstance(self.mean_module, (EWMAMean, DEWMAMean, TEWMAMean)):
            grad_flags = [True, False, False, False]
        else:
            grad_flags = [True, True, False, False, False]

        self.likelihood.raw_noise.data = torch.tensor([1e-5]).to(x.device)
        if vol_lh is not None:
            self.vol_lh = vol_lh.to(x.device)
        if vol_model is not None:
            self.vol_model = vol_model.to(x.device)

        for idx, p in enumerate(self.parameters()):
            p.requires_grad = grad_flags[idx]

        self.train()
        optimizer = torch.optim.Adam([{'params': self.parameters()}], lr=0.1)
        mll = ExactMarginalLogLikelihood(self.likelihood, self)

        print_every = 50
        for i in range(data_mod_iters):
            optimizer.zero_grad()
            output = self(x)
            loss = -mll(output, self.train_y)
            loss.backward()
            if display:
                if i % print_every == 0:
                    print('Iter %d/%d - Loss: %.3f' % (i + 1, data_mod_iters, loss.item()))
            optimizer.step()

    def Forecast(self, test_x, nsample=50, return_vol=False, mean_revert=False, theta=0.05, **rollout_kw):
        if self.vol_model is not None:
            self.vol_model.eval()
        self.eval()
        prices = torch.cat((self.train_targets[..., :1], self.train_targets.squeeze())).exp()   # Rollouts drops [0]
        samples = Rollouts(self.train_inputs[0].squeeze(), prices, test_x, self, nsample=nsample,
                           theta=theta if mean_revert else None, **rollout_kw)
        return samples
