"""Fit of the baseline GPs (the role of voltron/train_utils.py:146-190) on volt_amd's loop driver."""
from volt_amd import gp
from volt_amd.gp import GaussianLikelihood
from volt_amd.train_utils import LR_DATA, _fit_exact, _set_mean

from .models import MaternGP, SMGP


def TrainBasicModel(train_x, train_y, train_iters=1000, printing=False, model_type="matern", num_mixtures=10,
                    mean_func="loglinear"):
    log_y = train_y.log()
    lh = GaussianLikelihood()
    model = MaternGP(train_x, log_y, lh) if model_type == "matern" else SMGP(train_x, log_y, lh, num_mixtures)
    if mean_func == "loglinear":
        _set_mean(model, "loglinear", train_x, log_y)
        model.mean_module.register_prior("slope_prior", gp.NormalPrior(0, 0.1), 'weights')
    lh.raw_noise.data.fill_(1e-5)
    model = model.to(train_x.device)
    _fit_exact(model, lh, train_x, log_y, list(model.parameters()), LR_DATA, train_iters, printing)
    return model, lh
