"""Training loops on the HIP path -- the ``voltron.train_utils`` surface for SURVEY 8 rows a5 (the exact-GP data
model: TrainVoltMagpieModel :192-257, TrainDataModel :98-144), (f)1 (TrainVolModel :69-95) and (f)4 (LearnGPCV :15-67).

Signatures, defaults, the set of trained parameters and the arithmetic of one iteration are the reference's; the code
is organised differently: ONE loop driver (``_run_iterations``), ONE table of mean functions (``_MEANS``) and ONE
exact-GP fit (``_fit_exact``) serve every entry point, and which parameters train is stated by role (the likelihood's
noise and the mean module's own parameters; the vol forecaster stays frozen) -- the effect of the reference's positional
``grad_flags`` (:199-227).  ``mll`` is volt_amd.gp.ExactMarginalLogLikelihood: one fused HIP step with an analytic
backward.  ``TrainVoltMagpieBatch`` / ``TrainVolModelBatch`` fit B independent series in one batched model (the
reference loops over tickers in Python, experiments/stocks/ForecastGenerator.py:27-41), optionally sharded over ranks
with one all-reduce of the summed loss.
"""
import os
import warnings

import torch

from . import gp
from .gp import ExactMarginalLogLikelihood, GaussianLikelihood
from .means import DEWMAMean, EWMAMean, LogLinearMean, MeanRevertingEMAMean, TEWMAMean
from .models import VoltMagpie, VoltronGP

LR_DATA, LR_VOL, LR_GPCV = 0.1, 0.01, 0.01          # train_utils.py:238 / :81 / :43
PRINT_EVERY = 50
CHECK_EVERY = 25                                     # deferred loops: one host read of the info flags per this many iterations


# ------------------------------------------------------------------------------------------------ mean functions
def _loglinear(x, log_y, k, theta, bs):
    m = LogLinearMean(1, batch_shape=bs)
    m.initialize_from_data(x, log_y)                 # train_utils.py:102-103, :215-216
    return m


# name -> constructor(train_x, log_y, k, theta, batch_shape).  The moving-average family has no parameters; constant /
# loglinear / linear means train theirs together with the noise (the True entries of the reference's grad_flags).
_MEANS = {
    "ewma": lambda x, y, k, th, bs: EWMAMean(x, y, k),
    "dewma": lambda x, y, k, th, bs: DEWMAMean(x, y, k),
    "tewma": lambda x, y, k, th, bs: TEWMAMean(x, y, k),
    "meanrevert": lambda x, y, k, th, bs: MeanRevertingEMAMean(x, y, k, th),
    "constant": lambda x, y, k, th, bs: gp.ConstantMean(batch_shape=bs),
    "loglinear": _loglinear,
    "linear": lambda x, y, k, th, bs: gp.LinearMean(1, batch_shape=bs),
}


def _set_mean(model, mean_func, train_x, log_y, k=25, theta=0.5, batch_shape=torch.Size()):
    name = mean_func.lower()
    if name not in _MEANS:
        raise ValueError(f"unknown mean_func {mean_func!r}: one of {sorted(_MEANS)}")
    model.mean_module = _MEANS[name](train_x, log_y, k, theta, batch_shape).to(train_x.device)


def _train_noise_and_mean(model, lh, noise0=1e-5):
    """raw_noise starts at 1e-5 (:222); only it and the mean module's parameters receive gradients."""
    with torch.no_grad():
        lh.raw_noise.fill_(noise0)
    for p in model.parameters():
        p.requires_grad = False
    trainable = [lh.raw_noise] + list(model.mean_module.parameters())
    for p in trainable:
        p.requires_grad = True
    return trainable


# ------------------------------------------------------------------------------------------------ the loop driver
def _adam(params, lr, graph):
    """torch.optim.Adam as the reference builds it; for graph-captured loops the same update as two launches with the
    step count on the device (optim.FusedAdam: torch's capturable Adam is 13 launches per step and the iteration is
    launch-bound).  VOLT_TORCH_ADAM=1 keeps torch's in both."""
    if graph and not os.environ.get("VOLT_TORCH_ADAM"):
        from .optim import FusedAdam
        return FusedAdam(params, lr=lr)
    return torch.optim.Adam(params, lr=lr, capturable=bool(graph))


class _Snapshot:
    """Values of the parameters and of the optimiser's state tensors at one iteration; restored IN PLACE (a captured
    graph and the optimiser keep their addresses)."""

    def __init__(self, optimizer, it):
        self.opt = optimizer
        self.params = [p for g in optimizer.param_groups for p in g["params"]]
        self.take(it)

    def _state_tensors(self):
        out = []
        for p in self.params:
            st = self.opt.state.get(p, {})
            out += [v for _, v in sorted(st.items()) if torch.is_tensor(v)]
        extra = getattr(self.opt, "extra_state_tensors", None)
        return out + (extra() if extra else [])

    def take(self, it):
        self.it = it
        self.live = [p.data for p in self.params] + self._state_tensors()
        self.saved = [t.clone() for t in self.live]

    def restore(self):
        with torch.no_grad():
            for t, s in zip(self.live, self.saved):
                t.copy_(s)


def _run_iterations(iteration, optimizer, train_iters, printing, graph=False, scale=1.0, warm=3, defer=False, agree=None):
    """``train_iters`` times: zero_grad -> ``iteration()`` (forward + backward, returns the loss) -> print every 50th
    -> optimizer.step() -- the body of train_utils.py:243-254 and its siblings.

    Plain (graph=False, defer=False): eagerly, with the factorisation's ``info`` read back every step (a device
    synchronisation) so that gpytorch's jitter ladder can run at once.
    defer=True: the read-back moves to one check per CHECK_EVERY iterations (gp.deferred_checks); when a check finds a
    failed factorisation the parameters and the optimiser state return to the last clean snapshot and those
    iterations are replayed with the per-step check -- the trajectory is the plain loop's, without its host round trips.
    graph=True: after `warm` eager iterations ONE iteration (every HIP launch of the step, the torch glue, the Adam
    update) is captured into a hipGraph and replayed; a failed factorisation inside the replays is answered the same
    way: restore the post-warm-up snapshot and finish eagerly with the ladder.
    ``agree`` (distributed loops): maps this rank's "a factorisation failed" to the job's (an all-reduce MAX), so that
    EVERY rank replays the same stretch -- a replayed iteration issues the step's collective again, and a rank-local
    decision would pair iterations up with the wrong partner and leave an unmatched all-reduce at the end."""
    loss = None

    def any_bad():
        bad = chk.any_bad()
        return agree(bad) if agree is not None else bad

    def one(i):
        nonlocal loss
        optimizer.zero_grad(set_to_none=True)
        loss = iteration()
        if printing and i % PRINT_EVERY == 0:
            print('Iter %d/%d - Loss: %.3f' % (i + 1, train_iters, loss.item() * scale))
        optimizer.step()

    if (not graph and not defer) or train_iters <= warm + 1:
        for i in range(train_iters):
            one(i)
        return loss

    with gp.deferred_checks(immediate=True) as chk:
        # warm-up with the per-step check: creates the optimiser state, sizes the deferred accumulators
        side = torch.cuda.Stream() if graph else None
        if graph:
            side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side) if graph else _null():
            for i in range(warm):
                one(i)
        if graph:
            torch.cuda.current_stream().wait_stream(side)
        snap = _Snapshot(optimizer, warm)
        chk.immediate = False

        def replay_eagerly(lo, hi):
            snap.restore()
            chk.clear()
            chk.immediate = True
            for j in range(lo, hi):
                one(j)
            chk.immediate = False

        if graph:
            loss = None                                               # let the warm-up's autograd graph die before capture
            optimizer.zero_grad(set_to_none=True)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                static_loss = iteration()
                optimizer.step()
            for i in range(warm, train_iters):                        # capturing records an iteration, it does not run it
                g.replay()
                if printing and i % PRINT_EVERY == 0:
                    print('Iter %d/%d - Loss: %.3f' % (i + 1, train_iters, static_loss.item() * scale))
            loss = static_loss
            if any_bad():
                warnings.warn("a factorisation failed inside the captured loop: rerunning it eagerly with the jitter ladder",
                              gp.NumericalWarning)
                replay_eagerly(warm, train_iters)                     # (`one` rebinds `loss`: what is returned is the eager loop's)
            return loss
        i = warm
        while i < train_iters:
            hi = min(i + CHECK_EVERY, train_iters)
            for j in range(i, hi):
                one(j)
            if any_bad():                                             # the only host read of this stretch
                replay_eagerly(snap.it, hi)
            if hi < train_iters:
                snap.take(hi)
            i = hi
    return loss


class _null:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


CAPTURE_BELOW_MS = 4.0        # estimated step time under which the per-iteration host work (~0.3 ms of launches, autograd, Adam) is worth removing


def _capture_pays(target, n3_coeff=2.0 / 3.0):
    """Is an iteration over ``target`` ([.., N] log-prices) launch-bound?  The step is n3_coeff N^3 flop per series (2/3: the
    MLL + gradient step; 5/3: the GPCV ELBO step, csrc/gpcv.hip -- its own cost model) at the
    ~110 TFLOP/s the MLL step sustains; an eager iteration adds ~0.3 ms of host work (torch glue, autograd, Adam) that a
    captured one does not pay: 1 x 399 runs 0.50 ms eager / 0.14 captured, 64 x 4096 22.1 / 22.0.  Since round 5 the step
    of every shape is ONE launch per batch (or a handful), so capturing costs nothing -- the gate is only about whether
    it buys anything measurable."""
    n = target.shape[-1]
    batch = target.numel() // max(n, 1)
    npad = (n + 127) // 128 * 128
    return batch * n3_coeff * npad ** 3 / 110e12 * 1e3 < CAPTURE_BELOW_MS


def _auto_graph(graph, target, distributed=False, n3_coeff=2.0 / 3.0):
    """``graph=None`` (the default of every training loop here, as no reference call site passes it --
    voltron/train_utils.py:15,69,98,192): capture where it pays, on a CUDA device, outside an ongoing capture, single
    process.  An explicit True / False is honoured.
    NOTE what capturing changes besides the launches: the optimiser is `optim.FusedAdam` (the reference's Adam update as two
    launches with the step count on the device; `_adam`) instead of `torch.optim.Adam` -- the same arithmetic in fp32, checked
    against torch's in tests/test_gpu_api.py -- and a failed factorisation is answered after the fact: the loop restores the
    post-warm-up snapshot and finishes eagerly with gpytorch's jitter ladder (the loss it returns is then the eager one)."""
    if graph is not None:
        return bool(graph)
    if distributed or not target.is_cuda or torch.cuda.is_current_stream_capturing():
        return False
    return _capture_pays(target, n3_coeff)


def _fit_exact(model, lh, train_x, target, params, lr, train_iters, printing, graph=False, defer=False, batched=False,
               post_backward=None, scale=1.0, agree=None):
    """Adam on -mll(model(train_x), target); a batched model's per-series losses are summed for ONE backward (the
    series are independent and Adam is elementwise, so every series gets its own loop's update)."""
    model.train()
    lh.train()
    graph = bool(graph)      # (None is resolved by the public entry points: only THEIR models' steps are known to be capturable)
    optimizer = _adam([{'params': params}], lr, graph)
    mll = ExactMarginalLogLikelihood(lh, model)
    last = {}

    def iteration():
        losses = -mll(model(train_x), target)
        loss = losses.sum() if batched else losses
        loss.backward()
        last["losses"] = losses.detach()
        if post_backward is not None:
            return post_backward(loss)
        return loss

    _run_iterations(iteration, optimizer, train_iters, printing, graph, scale=scale, defer=defer, agree=agree)
    return last.get("losses")


# ------------------------------------------------------------------------------------------------ (f)4: GPCV
def FitGPCV(train_x, train_y, train_iters=1000, printing=False, kernel="bm", graph=None):
    """The fit inside LearnGPCV (train_utils.py:15-58), returning what it builds: (model, likelihood, losses)."""
    from .kernels import BMKernel, FBMKernel
    from .likelihoods import VolatilityGaussianLikelihood
    from .models import SingleTaskVariationalGP
    from .variational import VariationalELBO, num_gauss_hermite_locs
    dt = train_x[1] - train_x[0]
    yy = (train_y[..., 1:] - train_y[..., :-1]) / train_y[..., :-1] / dt ** 0.5        # scaled returns, :16-18
    kw = {"batch_shape": yy.shape[:-1]} if yy.ndim > 1 else {}
    likelihood = VolatilityGaussianLikelihood(param="exp")
    covar_module = {"bm": BMKernel, "fbm": FBMKernel}[kernel](**kw)
    model = SingleTaskVariationalGP(init_points=train_x.view(-1, 1), likelihood=likelihood, use_piv_chol_init=False,
                                    mean_module=gp.ConstantMean(**kw), covar_module=covar_module,
                                    learn_inducing_locations=False, use_whitened_var_strat=False)
    model.initialize_variational_parameters(likelihood, train_x, y=yy)
    model.train()
    likelihood.train()
    graph = bool(graph)                                  # (this entry returns the per-iteration losses: captured only on request)
    optimizer = _adam([{"params": model.parameters()}], LR_GPCV, graph)
    elbo = VariationalELBO(likelihood, model, yy.shape[-1], combine_terms=True)
    losses = []

    def iteration():                                     # :50-54
        with num_gauss_hermite_locs(75):
            loss = -elbo(model(train_x), yy)
            if not graph:
                losses.append(loss.detach())
            loss = loss.sum() if loss.ndim else loss     # independent series: one backward for all of them
            loss.backward()
        return loss

    last = _run_iterations(iteration, optimizer, train_iters, printing, graph)
    if graph and last is not None:
        losses.append(last.detach())
    model.eval()
    likelihood.eval()
    return model, likelihood, losses


def LearnGPCV(train_x, train_y, train_iters=1000, printing=False, early_stopping=False, kernel="bm", graph=None):
    """voltron/train_utils.py:15-67: the volatility path of a price series from a variational GP (BM or FBM prior over
    log-vol, ``y | f ~ N(0, exp f)``) fitted to the scaled returns; one HIP ELBO step per iteration.
    train_y [N+1] prices -> pred_scale [N]; train_y [T,N+1] fits T series at once (batched parameters)."""
    graph = _auto_graph(graph, train_y[..., 1:], n3_coeff=5.0 / 3.0)   # (only the fitted scale is returned: nothing per-iteration is lost)
    model, likelihood, _ = FitGPCV(train_x, train_y, train_iters=train_iters, printing=printing, kernel=kernel, graph=graph)
    return likelihood(model(train_x), return_gaussian=False).scale.mean(0).detach()      # :60-67


# ------------------------------------------------------------------------------------------------ (f)1: vol forecaster
def _vol_model(train_x, vol_path, kernel, batch_shape):
    from .models import BMGP
    vol_lh = GaussianLikelihood(batch_shape=batch_shape).to(train_x.device)
    # (the reference's `vol_lh.noise.data = 1e-2` at :71 writes to a temporary: the noise starts at softplus(0) + 1e-4)
    return BMGP(train_x, vol_path.log(), vol_lh, kernel=kernel).to(train_x.device), vol_lh


def TrainVolModel(train_x, vol_path, train_iters=1000, printing=False, kernel="bm", graph=None):
    """voltron/train_utils.py:69-95: the Brownian-motion GP over log-vol that supplies pred_vol to Rollouts.  The MLL
    and its gradient wrt the kernel's `vol` and the noise run on the HIP step (K = vol * min(x,x') keeps d mll / d vol
    in closed form, gp._ExactMLL)."""
    vol_model, vol_lh = _vol_model(train_x, vol_path, kernel, torch.Size())
    _fit_exact(vol_model, vol_lh, train_x, vol_path.log(), list(vol_model.parameters()), LR_VOL, train_iters, printing,
               _auto_graph(graph, vol_path))
    return vol_model, vol_lh


def TrainVolModelBatch(train_x, vol_path, train_iters=1000, printing=False, kernel="bm", graph=None):
    """TrainVolModel for T series at once: vol_path [T,N] -> one batched BMGP (per-series kernel parameter and noise)."""
    T = vol_path.shape[0]
    vol_model, vol_lh = _vol_model(train_x, vol_path, kernel, torch.Size([T]))
    graph = _auto_graph(graph, vol_path)
    _fit_exact(vol_model, vol_lh, train_x, vol_path.log(), list(vol_model.parameters()), LR_VOL, train_iters, printing, graph,
               defer=not graph, batched=True, scale=1.0 / T)
    return vol_model, vol_lh


# ------------------------------------------------------------------------------------------------ a5: the data model
def _attach_vol(model, vol_model, vol_lh, dev):
    if vol_lh is not None:
        model.vol_lh = vol_lh.to(dev)
    if vol_model is not None:
        model.vol_model = vol_model.to(dev)


def TrainDataModel(train_x, train_y, vol_model, vol_lh, vol_path, train_iters=1000, printing=False, graph=None):
    """voltron/train_utils.py:98-144: VoltronGP with a log-linear mean; noise, slope and intercept train."""
    dev = train_x.device
    log_y = train_y.log()
    lh = GaussianLikelihood().to(dev)
    model = VoltronGP(train_x, log_y, lh, vol_path)
    _set_mean(model, "loglinear", train_x, log_y)
    _attach_vol(model, vol_model, vol_lh, dev)
    params = _train_noise_and_mean(model, lh)
    _fit_exact(model, lh, train_x, log_y, params, LR_DATA, train_iters, printing, _auto_graph(graph, log_y))
    return model, lh


def TrainVoltMagpieModel(train_x, train_y, vol_model, vol_lh, vol_path, train_iters=1000, printing=False, k=25,
                         theta=0.5, mean_func="ewma", graph=None):
    """voltron/train_utils.py:192-257: VoltMagpie with the chosen mean; the noise (and a constant / (log)linear mean's
    parameters) train, the vol forecaster rides along frozen."""
    dev = train_x.device
    log_y = train_y.log()
    lh = GaussianLikelihood().to(dev)
    model = VoltMagpie(train_x, log_y, lh, vol_path, k=k).to(dev)
    if mean_func.lower() != "ewma":                                   # VoltMagpie's own mean is the EWMA
        _set_mean(model, mean_func, train_x, log_y, k, theta)
    _attach_vol(model, vol_model, vol_lh, dev)
    params = _train_noise_and_mean(model, lh)
    _fit_exact(model, lh, train_x, log_y, params, LR_DATA, train_iters, printing, _auto_graph(graph, log_y))
    return model, lh


def TrainVoltMagpieBatch(train_x, train_y, vol_path, train_iters=1000, k=25, printing=False, process_group=None,
                         shared_noise=False, mean_func="ewma", theta=0.5, graph=None, defer=True, reduce_across_ranks=True):
    """B independent series in one batched model (train_y [B,N] raw prices[1:], vol_path [B,N]).  Per-series raw_noise
    by default (each series is its own GP, as in the reference's loop over tickers); ``shared_noise`` ties one
    likelihood across series AND ranks, whose gradient is then all-reduced.  ``mean_func`` as in
    TrainVoltMagpieModel; constant / loglinear / linear means get one parameter set PER SERIES.  The per-step ``info``
    read-back is deferred (``defer``, see _run_iterations; under torch.distributed the "replay?" decision is taken
    collectively); ``graph=True`` captures the iteration (single process only: with ranks it falls back to the eager loop
    with a warning).  ``reduce_across_ranks=False`` keeps the fit rank-local even when torch.distributed is up -- no
    collective at all -- for callers whose ranks run DIFFERENT numbers of fits (the sharded forecast drivers).
    With ``reduce_across_ranks`` EVERY rank of the group must call this with the same ``train_iters``, ``defer`` and ``graph``
    (each stretch of the loop ends in a collective "replay?" decision, and a rank whose jitter ladder is exhausted raises
    only after that decision has been taken by all -- see ``agree``): checked once, up front, with an all-gather.
    Returns (model, likelihood, last per-series losses)."""
    from . import distributed as vdist
    B = train_y.shape[0]
    dev = train_x.device
    log_y = train_y.log()
    lh = GaussianLikelihood(batch_shape=torch.Size() if shared_noise else torch.Size([B])).to(dev)
    model = VoltMagpie(train_x, log_y, lh, vol_path, k=k).to(dev)
    if mean_func.lower() != "ewma":
        _set_mean(model, mean_func, train_x, log_y, k, theta, torch.Size([B]))
    params = _train_noise_and_mean(model, lh)
    distributed = (reduce_across_ranks and vdist._dist() is not None
                   and vdist._dist().get_world_size(process_group) > 1)
    if shared_noise and not reduce_across_ranks and vdist._dist() is not None and vdist._dist().get_world_size(process_group) > 1:
        raise ValueError("TrainVoltMagpieBatch: shared_noise ties the likelihood across ranks and needs reduce_across_ranks=True")
    if graph and distributed:
        warnings.warn("TrainVoltMagpieBatch: graph=True is for single-process runs (the all-reduce stays eager); running the "
                      "eager loop with the deferred check", RuntimeWarning, stacklevel=2)
        graph = False
    graph = _auto_graph(graph, log_y, distributed)
    count = torch.tensor(float(B), device=dev)

    def reduce(loss):                                     # the path's one collective: summed loss (and a shared gradient)
        if not distributed:
            return loss.detach() / B
        total = vdist.all_reduce_scalars(torch.stack([loss.detach(), count]), process_group)
        if shared_noise:
            vdist.all_reduce_(lh.raw_noise.grad, process_group)
        return total[0] / total[1]

    def agree(bad):                                       # every rank replays, or none does (see _run_iterations)
        flag = torch.tensor([1.0 if bad else 0.0], device=dev)
        vdist._dist().all_reduce(flag, op=vdist._dist().ReduceOp.MAX, group=process_group)
        return bool(flag.item() > 0)

    if distributed:                                       # the loop arguments the collectives depend on must agree
        mine = torch.tensor([float(train_iters), float(bool(defer)), float(bool(graph))], device=dev)
        allv = [torch.zeros_like(mine) for _ in range(vdist._dist().get_world_size(process_group))]
        vdist._dist().all_gather(allv, mine, group=process_group)
        if any(not torch.equal(v, mine) for v in allv):
            raise ValueError("TrainVoltMagpieBatch: train_iters / defer / graph differ across the ranks of the group: "
                             f"{[v.tolist() for v in allv]}")

    losses = _fit_exact(model, lh, train_x, log_y, params, LR_DATA, train_iters, printing, graph, defer=defer and not graph,
                        batched=True, post_backward=reduce, agree=agree if distributed else None)
    return model, lh, losses                                          # None for train_iters = 0 (GPGenerator.py:89-92)


def __getattr__(name):                                   # TrainBasicModel (voltron/train_utils.py:146-189): out of scope, but
    if name == "TrainBasicModel":                        # experiments/stocks/GenerateMultiMeanPreds.py:18 imports it
        from ._out_of_scope import resolve
        return resolve(name)
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
