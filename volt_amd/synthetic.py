"""Synthetic series for tests and benches.

The recipe is the SDE of the reference's walkthrough notebook (example.ipynb cells 2-3):
``F_t = F_{t-1} + V_{t-1} F_{t-1}^beta dW_t``, ``V_t = V_{t-1} + alpha V_{t-1} dZ_t``,
``dZ = rho dW + sqrt(1-rho^2) dW'``, F0=10, V0=0.2, beta=0.9, rho=-0.2, base seed 2019.
The notebook uses T=1 with 400 steps; here dt is the experiments' trading-day step
(experiments/stocks/GenerateMultiMeanPreds.py:75) and the vol-of-vol is scaled by 1/sqrt(T) so
that alpha^2 T keeps the notebook's value at every N (otherwise V collapses to 0 over 16 years).
Prices are floored at 1e-3 and |V| at 1e-3 so logs stay finite.
"""
from __future__ import annotations

import numpy as np

DT_STOCKS = 1.0 / 252     # GenerateMultiMeanPreds.py:75
DT_WIND = 1.0 / 365       # experiments/weather/GPGenerator.py:39


def sde_series(n: int, seed: int = 2019, dt: float = DT_STOCKS):
    """Returns (prices [n+1] fp32, vol_path [n] fp32).  y = log(prices[1:]) is what the data
    model trains on (train_utils.py:195-197 with train_y[1:])."""
    rng = np.random.RandomState(seed)
    steps = n + 1
    T = steps * dt
    F0, V0, alpha, beta, rho = 10.0, 0.2, 1.25 / np.sqrt(T), 0.9, -0.2
    dW = rng.normal(0, np.sqrt(dt), steps)
    dZ = rho * dW + np.sqrt(1 - rho ** 2) * rng.normal(0, np.sqrt(dt), steps)
    F = np.zeros(steps)
    V = np.zeros(steps)
    F[0], V[0] = F0, V0
    for t in range(1, steps):
        F[t] = max(F[t - 1] + V[t - 1] * F[t - 1] ** beta * dW[t], 1e-3)
        V[t] = V[t - 1] + alpha * V[t - 1] * dZ[t]
    vol = np.maximum(np.abs(V[1:]), 1e-3)
    return F.astype(np.float32), vol.astype(np.float32)


def sde_batch(batch: int, n: int, seed: int = 2019, dt: float = DT_STOCKS, first: int = 0):
    """``batch`` independent series; series i uses seed + first + i, so any rank's
    shard can be generated without the others.  Returns x [n], prices [B,n+1], vol [B,n]."""
    F = np.zeros((batch, n + 1), dtype=np.float32)
    V = np.zeros((batch, n), dtype=np.float32)
    for i in range(batch):
        F[i], V[i] = sde_series(n, seed + first + i, dt)
    x = (np.arange(n) * dt).astype(np.float32)
    return x, F, V


def rollout_inputs(vol_last: np.ndarray, nsample: int, horizon: int, seed: int = 2019, step_sd: float = 0.05):
    """pred_vol [.., S, H] (exp of a seeded Brownian log-vol path continuing the last train
    vol) and z [.., S, H] standard normals, both passed IN to rollouts so CPU and GPU consume
    identical numbers."""
    rng = np.random.RandomState(seed)
    vol_last = np.asarray(vol_last, dtype=np.float64)
    shape = vol_last.shape + (nsample, horizon)
    logv = np.log(vol_last)[..., None, None] + np.cumsum(rng.normal(0, step_sd, shape), axis=-1)
    z = rng.normal(0, 1, shape)
    return np.exp(logv).astype(np.float32), z.astype(np.float32)
