"""Training loops of the exact-GP data model -- drop-in for the two ★ loops of
voltron/train_utils.py (TrainDataModel :98-144, TrainVoltMagpieModel :192-257).

The loop bodies are the reference's, statement for statement: ``optimizer.zero_grad(); output =
voltron(train_x); loss = -mll(output, y); loss.backward(); optimizer.step()``.  What changes is
what runs underneath: ``mll`` is volt_amd.gp.ExactMarginalLogLikelihood, one fused HIP step with
an analytic backward.  LearnGPCV (:15-67, the variational GPCV stage) and TrainVolModel (:69-95) are here
too, on the same library, and so is TrainBasicModel (:146-190, Matern / spectral-mixture baselines: dense
d mll / d K from the HIP step, kernel derivatives by autograd).

``TrainVoltMagpieBatch`` is an addition for the multi-series case the reference only loops over
in Python (experiments/stocks/ForecastGenerator.py:27-41): B independent series in one batched
step, optionally sharded over ranks with one all-reduce of the summed loss.
"""
import os

import torch

from . import gp
from .gp import ExactMarginalLogLikelihood, GaussianLikelihood


def _adam(params, lr, graph):
    """torch.optim.Adam as the reference builds it (eager loops); for the graph-captured loops the same update as two
    launches with the step count on the device (optim.FusedAdam: torch's capturable Adam is 13 launches per step and
    the iteration is launch-bound).  VOLT_TORCH_ADAM=1 keeps torch's in both."""
    if graph and not os.environ.get("VOLT_TORCH_ADAM"):
        from .optim import FusedAdam
        return FusedAdam(params, lr=lr)
    return torch.optim.Adam(params, lr=lr, capturable=bool(graph))


def _run_iterations(iteration, optimizer, train_iters, printing, graph, scale=1.0, warm=3):
    """The reference's loop body ``optimizer.zero_grad(); output = model(x); loss = -mll(output, y); loss.backward();
    optimizer.step()`` (train_utils.py:243-254 and its siblings) run `train_iters` times.  ``iteration()`` does forward
    + backward and returns the loss.

    graph=False: eagerly, statement for statement.  graph=True: the first `warm` iterations eagerly on a side stream,
    then ONE iteration is captured into a hipGraph (torch.cuda.CUDAGraph: every HIP launch of the step, the torch glue
    and the capturable Adam update) and replayed -- at the reference's sizes (N = 399) an iteration is ~10 short launches
    plus ~0.4 ms of Python, i.e. launch-bound, and the replay removes the Python.  The per-step ``info`` read-back (a device
    synchronisation) is deferred to one check after the loop (gp.deferred_checks)."""
    print_every = 50
    if not graph or train_iters <= warm + 1:
        loss = None
        for i in range(train_iters):
            optimizer.zero_grad()
            loss = iteration()
            if printing and i % print_every == 0:
                print('Iter %d/%d - Loss: %.3f' % (i + 1, train_iters, loss.item() * scale))
            optimizer.step()
        return loss
    with gp.deferred_checks() as chk:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for i in range(warm):
                optimizer.zero_grad(set_to_none=True)
                iteration()                                       # (no reference to the loss kept: the autograd graph dies here)
                optimizer.step()
        torch.cuda.current_stream().wait_stream(side)
        optimizer.zero_grad(set_to_none=True)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            static_loss = iteration()
            optimizer.step()
        for i in range(warm, train_iters):                        # capturing records an iteration, it does not run it
            g.replay()
            if printing and i % print_every == 0:
                print('Iter %d/%d - Loss: %.3f' % (i + 1, train_iters, static_loss.item() * scale))
        chk.raise_if_bad()
    return static_loss
from .means import LogLinearMean, EWMAMean, DEWMAMean, TEWMAMean, MeanRevertingEMAMean
from .models import VoltronGP, VoltMagpie


def FitGPCV(train_x, train_y, train_iters=1000, printing=False, kernel="bm", graph=False):
    """The fit of LearnGPCV (train_utils.py:15-58) returning what it builds: (model, likelihood, losses)."""
    from .kernels import BMKernel, FBMKernel
    from .likelihoods import VolatilityGaussianLikelihood
    from .models import SingleTaskVariationalGP
    from .variational import VariationalELBO, num_gauss_hermite_locs
    dt = train_x[1] - train_x[0]
    scaled_returns = (train_y[..., 1:] - train_y[..., :-1]) / (train_y[..., :-1]) / (dt ** 0.5)
    yy = scaled_returns
    batch_shape = yy.shape[:-1]

    likelihood = VolatilityGaussianLikelihood(param="exp")
    kw = {"batch_shape": batch_shape} if len(batch_shape) else {}
    if kernel == "bm":
        covar_module = BMKernel(**kw)
    elif kernel == "fbm":
        covar_module = FBMKernel(**kw)
    model = SingleTaskVariationalGP(
        init_points=train_x.view(-1, 1), likelihood=likelihood, use_piv_chol_init=False,
        mean_module=gp.ConstantMean(**kw), covar_module=covar_module,
        learn_inducing_locations=False, use_whitened_var_strat=False
    )
    model.initialize_variational_parameters(likelihood, train_x, y=yy)

    model.train()
    likelihood.train()

    optimizer = _adam([
        {"params": model.parameters()},
    ], 0.01, graph)

    mll = VariationalELBO(likelihood, model, yy.shape[-1], combine_terms=True)

    losses = []

    def iteration():                                     # train_utils.py:50-54
        with num_gauss_hermite_locs(75):
            output = model(train_x)
            loss = -mll(output, yy)
            if not graph:
                losses.append(loss.detach())
            if loss.ndim:
                loss = loss.sum()                      # independent series: one backward for all of them
            loss.backward()
        return loss

    last = _run_iterations(iteration, optimizer, train_iters, printing, graph)
    if graph and last is not None:
        losses.append(last.detach())
    model.eval()
    likelihood.eval()
    return model, likelihood, losses


def LearnGPCV(train_x, train_y, train_iters=1000, printing=False, early_stopping=False, kernel="bm", graph=False):
    """voltron/train_utils.py:15-67 -- SURVEY 8(f) row 4: extract the volatility path from prices by fitting a
    variational GP (BM or FBM prior over log-vol, ``y | f ~ N(0, exp f)``) to the scaled returns.  Same statements
    as the reference; ``mll`` is volt_amd.variational.VariationalELBO, one HIP step per iteration.
    train_y [N+1] prices -> pred_scale [N]; train_y [T,N+1] fits T series at once (batched parameters)."""
    model, likelihood, _ = FitGPCV(train_x, train_y, train_iters=train_iters, printing=printing, kernel=kernel, graph=graph)
    predictive = model(train_x)
    pred_scale = likelihood(predictive, return_gaussian=False).scale.mean(0).detach()

    return pred_scale


def TrainVolModel(train_x, vol_path, train_iters=1000, printing=False, kernel="bm", graph=False):
    """voltron/train_utils.py:69-95 -- SURVEY 8(f) row 1: the Brownian-motion GP over log-vol that later
    supplies pred_vol to Rollouts.  Same loop; the MLL and its gradient wrt the kernel's `vol` and the
    noise run on the HIP path (K = vol * min(x,x') keeps d mll / d vol in closed form, gp._ExactMLL).
    Quirk kept: `vol_lh.noise.data = ...` (:71) assigns to a temporary in the reference and changes
    nothing, so the noise starts at softplus(0) + 1e-4."""
    from .models import BMGP
    vol_lh = GaussianLikelihood().to(train_x.device)
    vol_lh.noise.data = torch.tensor([1e-2])          # no-op, as in the reference
    vol_model = BMGP(train_x, vol_path.log(), vol_lh, kernel=kernel).to(train_x.device)

    optimizer = _adam([{'params': vol_model.parameters()}], 0.01, graph)
    mll = ExactMarginalLogLikelihood(vol_lh, vol_model)
    log_vol = vol_path.log()

    def iteration():                                     # train_utils.py:84-88
        output = vol_model(train_x)
        loss = -mll(output, log_vol)
        loss.backward()
        return loss

    _run_iterations(iteration, optimizer, train_iters, printing, graph)
    return vol_model, vol_lh


def TrainVolModelBatch(train_x, vol_path, train_iters=1000, printing=False, kernel="bm", graph=False):
    """TrainVolModel for T series at once: vol_path [T,N] -> one batched BMGP (per-series kernel parameter and noise).
    The series are independent, so the summed loss gives every series exactly the gradient its own TrainVolModel
    loop would (Adam is elementwise)."""
    from .models import BMGP
    T = vol_path.shape[0]
    vol_lh = GaussianLikelihood(batch_shape=torch.Size([T])).to(train_x.device)
    vol_model = BMGP(train_x, vol_path.log(), vol_lh, kernel=kernel).to(train_x.device)
    optimizer = _adam([{'params': vol_model.parameters()}], 0.01, graph)
    mll = ExactMarginalLogLikelihood(vol_lh, vol_model)
    log_vol = vol_path.log()

    def iteration():
        output = vol_model(train_x)
        loss = -mll(output, log_vol).sum()
        loss.backward()
        return loss

    _run_iterations(iteration, optimizer, train_iters, printing, graph, scale=1.0 / T)
    return vol_model, vol_lh


def TrainBasicModel(train_x, train_y, train_iters=1000, printing=False, model_type="matern", num_mixtures=10,
                    mean_func="loglinear"):
    """voltron/train_utils.py:146-190 -- SURVEY 8(f) row 2: the Matern / spectral-mixture baselines on log prices."""
    from .models import MaternGP, SMGP
    lh = GaussianLikelihood()

    if model_type == "matern":
        model = MaternGP(train_x, train_y.log(), lh)
    else:
        model = SMGP(train_x, train_y.log(), lh, num_mixtures)

    if mean_func == "loglinear":
        model.mean_module = LogLinearMean(1)
        model.mean_module.register_prior("slope_prior", gp.NormalPrior(0, 0.1), 'weights')
        model.mean_module.initialize_from_data(train_x, train_y.log())

    model.likelihood.raw_noise.data = torch.tensor([1e-5])
    model = model.to(train_x.device)
    model.train()
    lh.train()

    optimizer = torch.optim.Adam([{'params': model.parameters()}], lr=0.1)
    mll = ExactMarginalLogLikelihood(lh, model)
    print_every = 50
    for i in range(train_iters):
        optimizer.zero_grad()
        output = model(train_x)
        loss = -mll(output, train_y.log())
        loss.backward()
        if printing:
            if i % print_every == 0:
                print('Iter %d/%d - Loss: %.3f' % (i + 1, train_iters, loss.item()))
        optimizer.step()

    return model, lh


def TrainDataModel(train_x, train_y, vol_model, vol_lh, vol_path, train_iters=1000, printing=False, graph=False):
    voltron_lh = GaussianLikelihood().to(train_x.device)
    voltron = VoltronGP(train_x, train_y.log(), voltron_lh, vol_path)
    voltron.mean_module = LogLinearMean(1).to(train_x.device)
    voltron.mean_module.initialize_from_data(train_x, train_y.log())
    voltron.likelihood.raw_noise.data = torch.tensor([1e-5]).to(train_x.device)
    voltron.vol_lh = vol_lh
    voltron.vol_model = vol_model

    grad_flags = [True, True, True, False, False, False]

    for idx, p in enumerate(voltron.parameters()):
        p.requires_grad = grad_flags[idx]

    voltron.train()
    voltron_lh.train()

    optimizer = _adam([{'params': voltron.parameters()}], 0.1, graph)
    mll = ExactMarginalLogLikelihood(voltron_lh, voltron)
    log_y = train_y.log()

    def iteration():                                     # train_utils.py:245-250
        output = voltron(train_x)
        loss = -mll(output, log_y)
        loss.backward()
        return loss

    _run_iterations(iteration, optimizer, train_iters, printing, graph)
    return voltron, voltron_lh


def TrainVoltMagpieModel(train_x, train_y, vol_model, vol_lh, vol_path, train_iters=1000, printing=False, k=25,
                         theta=0.5, mean_func="ewma", graph=False):
    voltron_lh = GaussianLikelihood().to(train_x.device)
    voltron = VoltMagpie(train_x, train_y.log(), voltron_lh, vol_path, k=k).to(train_x.device)

    if mean_func.lower() in ["ewma", "dewma", "tewma", "meanrevert"]:
        # default voltmagpie is an ewma mean so we don't need to redefine anything
        grad_flags = [True, False, False, False]

        if mean_func.lower() == "dewma":
            voltron.mean_module = DEWMAMean(train_x, train_y.log(), k).to(train_x.device)
        elif mean_func.lower() == 'tewma':
            voltron.mean_module = TEWMAMean(train_x, train_y.log(), k).to(train_x.device)
        elif mean_func.lower() == 'meanrevert':
            voltron.mean_module = MeanRevertingEMAMean(train_x, train_y.log(), k, theta).to(train_x.device)

    elif mean_func.lower() == 'constant':
        voltron.mean_module = gp.ConstantMean().to(train_x.device)
        grad_flags = [True, True, False, False, False]
    elif mean_func.lower() == 'loglinear':
        voltron.mean_module = LogLinearMean(1).to(train_x.device)
        voltron.mean_module.initialize_from_data(train_x, train_y.log())
        grad_flags = [True, True, True, False, False, False]
    elif mean_func.lower() == 'linear':
        voltron.mean_module = gp.LinearMean(1).to(train_x.device)
        grad_flags = [True, True, True, False, False, False]

    voltron.likelihood.raw_noise.data = torch.tensor([1e-5]).to(train_x.device)
    if vol_lh is not None:
        voltron.vol_lh = vol_lh.to(train_x.device)
    if vol_model is not None:
        voltron.vol_model = vol_model.to(train_x.device)

    for idx, p in enumerate(voltron.parameters()):
        p.requires_grad = grad_flags[idx]

    voltron.train()
    voltron_lh.train()

    optimizer = _adam([{'params': voltron.parameters()}], 0.1, graph)
    mll = ExactMarginalLogLikelihood(voltron_lh, voltron)
    log_y = train_y.log()

    def iteration():                                     # train_utils.py:245-250
        output = voltron(train_x)
        loss = -mll(output, log_y)
        loss.backward()
        return loss

    _run_iterations(iteration, optimizer, train_iters, printing, graph)
    return voltron, voltron_lh


def TrainVoltMagpieBatch(train_x, train_y, vol_path, train_iters=1000, k=25, printing=False, process_group=None,
                         shared_noise=False, mean_func="ewma", theta=0.5):
    """B independent series in one batched model (train_y [B,N] raw prices[1:], vol_path [B,N]).
    Per-series raw_noise by default (each series is its own GP, as in the reference's Python loop over
    tickers); ``shared_noise`` ties one likelihood across series AND ranks, whose gradient is then
    all-reduced.  ``mean_func`` as in TrainVoltMagpieModel (train_utils.py:199-219): the EWMA family has
    no trainable parameters; 'constant' / 'loglinear' / 'linear' get one parameter set PER SERIES (batch_shape [B]),
    trained with the noise like the reference's grad_flags say.  Returns (model, likelihood, last per-series losses)."""
    from . import distributed as vdist
    B = train_y.shape[0]
    dev = train_x.device
    lh = GaussianLikelihood(batch_shape=torch.Size() if shared_noise else torch.Size([B])).to(dev)
    model = VoltMagpie(train_x, train_y.log(), lh, vol_path, k=k).to(dev)
    mf = mean_func.lower()
    trainable = []
    if mf == "dewma":
        model.mean_module = DEWMAMean(train_x, train_y.log(), k).to(dev)
    elif mf == "tewma":
        model.mean_module = TEWMAMean(train_x, train_y.log(), k).to(dev)
    elif mf == "meanrevert":
        model.mean_module = MeanRevertingEMAMean(train_x, train_y.log(), k, theta).to(dev)
    elif mf == "constant":
        model.mean_module = gp.ConstantMean(batch_shape=torch.Size([B])).to(dev)
    elif mf == "loglinear":
        model.mean_module = LogLinearMean(1, batch_shape=torch.Size([B])).to(dev)
        model.mean_module.initialize_from_data(train_x, train_y.log())
    elif mf == "linear":
        model.mean_module = gp.LinearMean(1, batch_shape=torch.Size([B])).to(dev)
    elif mf != "ewma":
        raise ValueError(f"unknown mean_func {mean_func!r}")
    lh.raw_noise.data.fill_(1e-5)
    for p in model.parameters():
        p.requires_grad = False
    lh.raw_noise.requires_grad = True
    if mf in ("constant", "loglinear", "linear"):
        trainable = list(model.mean_module.parameters())
        for p in trainable:
            p.requires_grad = True
    model.train()
    optimizer = torch.optim.Adam([lh.raw_noise] + trainable, lr=0.1)
    mll = ExactMarginalLogLikelihood(lh, model)
    losses = None
    for i in range(train_iters):
        optimizer.zero_grad()
        output = model(train_x)
        losses = -mll(output, train_y.log())
        loss = losses.sum()
        loss.backward()
        total = vdist.all_reduce_scalars(torch.stack([loss.detach(), torch.tensor(float(B), device=loss.device)]),
                                         process_group)
        if shared_noise:
            vdist.all_reduce_(lh.raw_noise.grad, process_group)
        if printing and i % 50 == 0:
            print('Iter %d/%d - Loss: %.3f' % (i + 1, train_iters, (total[0] / total[1]).item()))
        optimizer.step()
    return model, lh, (None if losses is None else losses.detach())      # train_iters = 0 (GPGenerator.py:89-92)
