"""Batched forecast drivers -- SURVEY 8(f) row 3: the sliding-window schedule and output format of
experiments/stocks/GenerateMultiMeanPreds.py:63-137 (and of the weather driver's volt/ewma branch,
experiments/weather/GPGenerator.py:20-112), with the reference's per-ticker / per-station Python ``for``
(ForecastGenerator.py:27-41, GPGenerator.py --stn_idx) replaced by one batched pass per window:

    window -> LearnGPCV for all tickers at once (batched variational fit, HIP ELBO step)          (:95-96)
           -> TrainVoltMagpieBatch (all tickers in one batched model, HIP MLL step)               (:99-103)
           -> TrainVolModelBatch + posterior samples of the vol forecasters (batched BM-GP, HIP factorisation)
           -> rollout engine for all tickers x paths in one launch                                (:105-107)
           -> torch.save(samples[S,H], "saved-outputs/<ticker>/<model>_<date>.pt")                (:128)

``vol_fn(train_x, prices [B,N+1]) -> vol [B,N]`` is the volatility-extraction stage: by default the reference's
LearnGPCV (train_utils.py:15-67) fitted for all tickers in one batch; ``realised_vol`` is a cheap non-reference
estimator kept for quick smoke runs.
"""
from __future__ import annotations

import os
import warnings

import torch

from . import rollout_engine
from .gp import NanError, NotPSDError, NumericalWarning
from .distributed import shard_range
from .rollout_utils import _posterior_draw
from .train_utils import LearnGPCV, TrainVoltMagpieBatch, TrainVolModelBatch

_MODES = {"ewma": 0, "dewma": 1, "tewma": 2}
_STANDARD = ("constant", "loglinear", "linear")


def realised_vol(train_x, prices, span=20, floor=1e-3):
    """Annualised EWMA of |log-returns| -- a stand-in for the GPCV scale (NOT the reference's estimator).
    prices [B, N+1] -> vol [B, N]."""
    dt = (train_x[1] - train_x[0]).item()
    r = (prices[:, 1:].log() - prices[:, :-1].log()).abs() / dt ** 0.5
    from . import ops
    v = ops.ewma(r, span)[:, 1:]
    return v.clamp_min(floor)


def _standard_mean_prediction(model, train_x, log_y, vol, test_x, pred_vol, z):
    """"VOLT + standard mean" (GenerateMultiMeanPreds.py:113-119): ONE joint draw over the whole horizon per path,
    GeneratePrediction(train_x, train_y, test_x, predvol, voltron) with predvol [S,H] -- rollout_utils.py:6-53 -- for
    every series of the batched model in turn (each is its own GP; the S systems of a series run batched on the
    device exactly as in the per-series function).  pred_vol, z [B,S,H] -> samples [B,S,H]."""
    B, S, H = pred_vol.shape
    N = train_x.numel()
    full_x = torch.cat((train_x, test_x))
    with torch.no_grad():
        m_tr = model.mean_module(train_x.unsqueeze(-1)).reshape(B, N)
        m_te = model.mean_module(test_x.unsqueeze(-1)).reshape(B, H)
    out = torch.empty(B, S, H, device=train_x.device)
    for b in range(B):
        full_vol = torch.cat((vol[b].unsqueeze(0).expand(S, N), pred_vol[b]), -1)        # :17-20
        draw = _posterior_draw(model.covar_module, full_x, full_vol, N, (log_y[b] - m_tr[b]).reshape(1, N, 1),
                               m_te[b].reshape(1, H, 1), z[b].reshape(S, H, 1), 1e-4)      # :26-53
        out[b] = draw.squeeze(-1)
    return out


def _window_pass(train_x, test_x, train_y, nsample, mean, k, gpcv_iters, vol_iters, data_iters, theta, vol_fn, generator,
                 graph, debug=None):
    """One window for the series in train_y [b, ntrain] (prices): GPCV -> data model -> vol forecasters -> rollouts,
    every stage for all b series at once.  Returns samples [b, S, H] on the device."""
    dev = train_y.device
    b, H = train_y.shape[0], test_x.numel()
    if vol_fn is None:
        vol = LearnGPCV(train_x, train_y, train_iters=gpcv_iters, graph=graph)           # all series at once
    else:
        vol = vol_fn(train_x, train_y)                                                   # [b, ntrain-1]
    # the shards are independent series and the ranks may run different numbers of fits (a failed window is redone series
    # by series on the rank that saw it): no collective in here
    model, lh, _ = TrainVoltMagpieBatch(train_x, train_y[:, 1:], vol, train_iters=data_iters, k=k, mean_func=mean,
                                        graph=graph, reduce_across_ranks=False)
    vmod, vlh = TrainVolModelBatch(train_x, vol, train_iters=vol_iters, graph=graph)
    vmod.eval()
    pred_vol = vmod(test_x).sample(torch.Size((nsample,))).exp().transpose(0, 1).contiguous().detach()   # [b,S,H]
    z = torch.randn(b, nsample, H, device=dev, generator=generator)
    latent = train_y.log().mean(-1) if theta is not None else None                       # rollout_utils.py:60-63
    if mean in _MODES:                                                                   # Rollouts, :110-112
        samples, info = rollout_engine.rollout_series(train_x, train_y[:, 1:].log(), vol.log(), test_x, pred_vol, z,
                                                      _MODES[mean], k, latent_mean=latent, theta=theta)
        if bool((info < 0).any()):                      # psd_safe_cholesky(pred_cov) would have raised for these paths (:46)
            raise NotPSDError("rollouts: predictive variance not positive after the jitter ladder")
    else:                                                                                # VOLT + standard mean, :113-119
        samples = _standard_mean_prediction(model, train_x, train_y[:, 1:].log(), vol, test_x, pred_vol, z)
    if debug is not None:
        debug.update(vol=vol, pred_vol=pred_vol, z=z, model=model, train_y=train_y)
    return samples.detach()                                                              # .detach(): :118


def _forecast_windows(names, series, end_idxs, ntrain, train_x, test_x, nsample, mean, k, gpcv_iters, vol_iters,
                      data_iters, theta, vol_fn, generator, save, path_fn, debug=None, graph=None):
    """One batched pass per window (series [B,T] prices; the window ending at index e trains on series[:, e-ntrain:e]).
    A numerical failure anywhere in the batched pass (NotPSDError / NanError after the jitter ladders) must not take the
    other series down with it: the window is then redone one series at a time, and a series that still fails gets NaN
    samples and a "Failed:" line -- what the reference's per-ticker try / except does
    (experiments/stocks/GenerateMultiMeanPreds.py:185-198).
    ``debug`` (a dict) receives the last window's intermediates (vol, pred_vol, z, model) for tests."""
    B = series.shape[0]
    H = test_x.numel()
    args = (nsample, mean, k, gpcv_iters, vol_iters, data_iters, theta, vol_fn, generator, graph)
    last = None
    for last_day in end_idxs:
        train_y = series[:, last_day - ntrain:last_day].float()                          # [B, ntrain] prices
        try:
            samples = _window_pass(train_x, test_x, train_y, *args, debug=debug)
        except (NotPSDError, NanError) as err:
            warnings.warn(f"window ending at {last_day}: the batched pass failed ({err}); redoing it series by series",
                          NumericalWarning)
            samples = torch.full((B, nsample, H), float("nan"), device=series.device)
            for b in range(B):
                try:
                    samples[b] = _window_pass(train_x, test_x, train_y[b:b + 1], *args)[0]
                except (NotPSDError, NanError):
                    print("Failed: ", names[b], mean, k)
        last = samples.cpu()
        if save:
            for b, name in enumerate(names):
                path = path_fn(name, last_day)
                os.makedirs(os.path.dirname(path), exist_ok=True)
                torch.save(last[b], path)
    return last


def GenerateStockPredictionsBatch(tickers, closes, dates=None, forecast_horizon=20, train_iters=400, nsample=1000,
                                  ntrain=400, mean="ewma", save=False, k=300, ntimes=-1, vol_fn=None,
                                  vol_iters=None, par_dir="./saved-outputs/", generator=None, debug=None, graph=None):
    """closes [B, T] prices for B tickers on a common calendar (device tensor).  Same window schedule,
    model name and file layout as GenerateStockPredictions (GenerateMultiMeanPreds.py:69-83,128); ``mean`` in
    ewma / dewma / tewma takes the Rollouts branch (:110-112), constant / loglinear / linear the "VOLT + standard
    mean" branch (:113-119: one multi-point GeneratePrediction per path).
    Under torch.distributed each rank takes a contiguous shard of the tickers.  Returns the samples of
    the last window, [B_local, nsample, forecast_horizon] on the CPU."""
    if mean not in _MODES and mean not in _STANDARD:
        raise ValueError(f"unknown mean {mean!r}: one of {sorted(_MODES) + list(_STANDARD)}")
    dev = closes.device
    lo, hi = shard_range(len(tickers))
    tickers, closes = list(tickers)[lo:hi], closes[lo:hi]
    T = closes.shape[1]
    if ntimes == -1:
        end_idxs = torch.arange(ntrain, T)
    else:
        end_idxs = torch.arange(ntrain, T, int((T - ntrain) / ntimes))
    dt = 1. / 252
    model_name = "volt_" + mean + str(k) + "_"
    vol_iters = train_iters if vol_iters is None else vol_iters
    train_x = torch.arange(ntrain - 1, device=dev) * dt                                  # :89
    test_x = torch.arange(forecast_horizon, device=dev) * dt + train_x[-1] + train_x[1]  # :90

    def path_fn(tckr, last_day):
        date = str(last_day) if dates is None else str(dates[last_day])
        return os.path.join(par_dir, tckr, model_name + date + ".pt")                    # :128
    return _forecast_windows(tickers, closes, end_idxs.tolist(), ntrain, train_x, test_x, nsample, mean, k,
                             train_iters, vol_iters, train_iters, None, vol_fn, generator, save, path_fn, debug, graph)


def GenerateWindPredictionsBatch(stations, data, forecast_horizon=100, ntrain=400, n_test_times=10, nsample=1000, k=400,
                                 theta=0.01, gpcv_iters=200, vol_iters=500, data_iters=0, save=False, vol_fn=None,
                                 par_dir="./saved-outputs/", generator=None, graph=None):
    """The ``--kernel volt --mean ewma`` branch of experiments/weather/GPGenerator.py:20-112 for B stations at once:
    data [B,T] wind speeds (missing = -99 -> 0, then +1 as at :47,55), dt = 1/365 (:38-41), the schedule of test
    windows of :33-34, GPCV 200 / vol model 500 / data model 0 iterations (:64-67,89-92), EWMA(k=400) mean and
    mean-reverting rollouts with theta = 0.01 (:96-102), files ``stn<idx>/volt_ema<k>_theta<theta>_<last_day>.pt``
    (:103-106).  Stations shard across ranks like tickers.  Returns the last window's samples [B_local,S,H] (CPU)."""
    dev = data.device
    lo, hi = shard_range(len(stations))
    stations, data = list(stations)[lo:hi], data[lo:hi].float()
    data = torch.where(data == -99.0, torch.zeros_like(data), data) + 1                  # :47,55
    ntime = data.shape[1]
    step = int((ntime - forecast_horizon - ntrain) / n_test_times)
    end_idxs = torch.arange(ntrain, ntime - forecast_horizon, step).tolist()            # :33-34
    train_x = torch.arange(ntrain - 1, device=dev).float() / 365                         # :38
    test_x = torch.arange(ntrain, ntrain + forecast_horizon, device=dev).float() / 365   # :41 (absolute day index, as written)

    def path_fn(stn, last_day):
        return os.path.join(par_dir, "stn" + str(stn), "volt_ema" + str(k) + "_theta" + str(theta) + "_" +
                            str(last_day) + ".pt")
    return _forecast_windows(stations, data, end_idxs, ntrain, train_x, test_x, nsample, "ewma", k, gpcv_iters,
                             vol_iters, data_iters, theta, vol_fn, generator, save, path_fn, None, graph)
