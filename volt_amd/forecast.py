"""Batched forecast driver -- SURVEY 8(f) row 3: the sliding-window schedule and output format of
experiments/stocks/GenerateMultiMeanPreds.py:63-137, with the reference's per-ticker Python ``for``
(ForecastGenerator.py:27-41) replaced by one batched pass per window:

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

import torch

from . import rollout_engine
from .distributed import shard_range
from .means import EWMAMean, DEWMAMean, TEWMAMean
from .train_utils import LearnGPCV, TrainVoltMagpieBatch, TrainVolModelBatch

_MODES = {"ewma": 0, "dewma": 1, "tewma": 2}


def realised_vol(train_x, prices, span=20, floor=1e-3):
    """Annualised EWMA of |log-returns| -- a stand-in for the GPCV scale (NOT the reference's estimator).
    prices [B, N+1] -> vol [B, N]."""
    dt = (train_x[1] - train_x[0]).item()
    r = (prices[:, 1:].log() - prices[:, :-1].log()).abs() / dt ** 0.5
    from . import ops
    v = ops.ewma(r, span)[:, 1:]
    return v.clamp_min(floor)


def GenerateStockPredictionsBatch(tickers, closes, dates=None, forecast_horizon=20, train_iters=400, nsample=1000,
                                  ntrain=400, mean="ewma", save=False, k=300, ntimes=-1, vol_fn=None,
                                  vol_iters=None, par_dir="./saved-outputs/", generator=None):
    """closes [B, T] prices for B tickers on a common calendar (device tensor).  Same window schedule,
    model name and file layout as GenerateStockPredictions (GenerateMultiMeanPreds.py:69-83,128).
    Under torch.distributed each rank takes a contiguous shard of the tickers.  Returns the samples of
    the last window, [B_local, nsample, forecast_horizon] on the CPU."""
    if mean not in _MODES:
        raise NotImplementedError("the batched driver covers the EWMA mean family (Rollouts path, :110-112)")
    dev = closes.device
    lo, hi = shard_range(len(tickers))
    tickers, closes = list(tickers)[lo:hi], closes[lo:hi]
    B, T = closes.shape
    if ntimes == -1:
        end_idxs = torch.arange(ntrain, T)
    else:
        end_idxs = torch.arange(ntrain, T, int((T - ntrain) / ntimes))
    dt = 1. / 252
    model_name = "volt_" + mean + str(k) + "_"
    vol_iters = train_iters if vol_iters is None else vol_iters
    last = None
    for last_day in end_idxs.tolist():
        date = str(last_day) if dates is None else str(dates[last_day])
        train_y = closes[:, last_day - ntrain:last_day].float()                       # [B, ntrain] prices
        train_x = torch.arange(ntrain - 1, device=dev) * dt                             # :89
        test_x = torch.arange(forecast_horizon, device=dev) * dt + train_x[-1] + train_x[1]     # :90
        if vol_fn is None:
            vol = LearnGPCV(train_x, train_y, train_iters=train_iters)                  # :95-96, all tickers at once
        else:
            vol = vol_fn(train_x, train_y)                                              # [B, ntrain-1]
        # data model: all tickers in one batched VoltMagpie (per-ticker noise), :104-108
        model, lh, _ = TrainVoltMagpieBatch(train_x, train_y[:, 1:], vol, train_iters=train_iters, k=k)
        if mean != "ewma":
            cls = {"dewma": DEWMAMean, "tewma": TEWMAMean}[mean]
            model.mean_module = cls(train_x, train_y[:, 1:].log(), k)
        # vol forecasters (BM-GP over log-vol, :102), all tickers in one batched model, and their posterior samples
        # (rollout_utils.py:66)
        vmod, vlh = TrainVolModelBatch(train_x, vol, train_iters=vol_iters)
        vmod.eval()
        pred_vol = vmod(test_x).sample(torch.Size((nsample,))).exp().transpose(0, 1).contiguous()   # [B,S,H]
        z = torch.randn(B, nsample, forecast_horizon, device=dev, generator=generator)
        samples, info = rollout_engine.rollout_series(train_x, train_y[:, 1:].log(), vol.log(), test_x, pred_vol, z,
                                                      _MODES[mean], k)
        last = samples.cpu()
        if save:
            os.makedirs(par_dir, exist_ok=True)
            for b, tckr in enumerate(tickers):
                savepath = os.path.join(par_dir, tckr)
                os.makedirs(savepath, exist_ok=True)
                torch.save(last[b], os.path.join(savepath, model_name + date + ".pt"))  # :128
    return last
