"""Posterior conditionals and sequential rollouts -- drop-in for voltron/rollout_utils.py.

``GeneratePrediction`` / ``Rollouts`` keep the reference's signatures, shapes, CPU-resident
``samples`` result and the in-place mutation of ``model`` between horizon steps
(rollout_utils.py:80-86; callers defend with copy.deepcopy, experiments/weather/GPGenerator.py:79).

Two engines compute the same conditional:

* ``engine="dense"`` -- the reference's algorithm line by line, on the device: at every horizon step
  fill S matrices of size (N+idx+1)^2, Cholesky-factor every train block, two solves, one draw
  (rollout_utils.py:26-48).  O(H S N^3); kept as the on-device restatement used by the parity tests.
* ``engine="bordered"`` (default) -- volt_rollout_* HIP kernels: the train block of every sample's
  matrix is identical (``train_stack_vol = log_vol_path.repeat(S,1)``, :72), so it is factored once
  and each sample only extends its own <= H x H bordered factor.  See DESIGN.md "Rollouts".

Extra keyword-only arguments (``pred_vol``, ``z``, ``engine``) let tests inject the vol-path
sample and the N(0,1) draws the reference takes from ``model.vol_model`` and ``torch.randn``.
"""
from __future__ import annotations

import os
import warnings

import torch

from . import ops
from .gp import NanError, NotPSDError, NumericalWarning, _safe_factor


def _chol_1x1(pc: torch.Tensor, jitter):
    """psd_safe_cholesky of a [...,1,1] matrix (rollout_utils.py:46): sqrt, with the jitter ladder."""
    if bool((pc > 0).all()):
        return pc.sqrt()
    if torch.isnan(pc).any():
        raise NanError("cholesky: NaN in the predictive covariance")
    jitter = 1e-6 if jitter is None else jitter
    for i in range(3):
        pj = pc + jitter * (10 ** i)
        if bool((pj > 0).all()):
            warnings.warn(f"A not p.d., added jitter of {jitter * 10 ** i:.1e} to the diagonal", NumericalWarning)
            return pj.sqrt()
    raise NotPSDError("predictive covariance not positive definite after adding jitter")


def _posterior_draw(covar_module, full_x, full_vol, idx_cut, train_diffs, test_mean_term, z, jitter,
                    latent_mean=None, theta=0.5):
    """rollout_utils.py:26-53 on the device.  full_x [N+T] or [S,N+T]; full_vol [S,N+T] (or [N+T]);
    train_diffs [S,N,1] (or broadcastable); test_mean_term broadcastable to [S,T,1]; z [S,T,n].
    Returns samples + pred_mean, shape [S,T,n]."""
    cov_mat = covar_module(full_x.unsqueeze(-1), full_vol.unsqueeze(-1)).evaluate()        # :26  (HIP fill)
    if cov_mat.ndim == 2:
        cov_mat = cov_mat.unsqueeze(0)
    S = cov_mat.shape[0]
    T = cov_mat.shape[-1] - idx_cut
    K_tr = cov_mat[..., :idx_cut, :idx_cut]                                                # :27
    K_te_tr = cov_mat[..., idx_cut:, :idx_cut]                                             # :28, transposed: rows = test points
    K_te = cov_mat[..., idx_cut:, idx_cut:]                                                # :29
    f, _ = _safe_factor(K_tr, jitter)                                                      # :35  (HIP potrf)
    td = train_diffs.reshape(-1, idx_cut) if train_diffs.ndim > 1 else train_diffs.reshape(1, idx_cut)
    td = td.expand(S, idx_cut)
    sol = ops.cholesky_solve(f, td)                                                        # :36
    # every product below runs on the library's own MFMA GEMM (volt_gemm_nt_f32), not on a vendor BLAS
    dt = cov_mat.dtype
    pred_mean = ops.gemm_nt(K_te_tr, sol.unsqueeze(1)).to(dt)                              # [S,T,1] = K_te,tr sol
    pred_mean = pred_mean + test_mean_term                                                 # :39
    if latent_mean is not None:
        pred_mean = pred_mean - theta * (pred_mean - latent_mean)                          # :41-42
    if T == 1:
        sol2 = ops.cholesky_solve(f, K_te_tr[..., 0, :])                                   # :44
        pred_cov = K_te - (K_te_tr[..., 0, :] * sol2).sum(-1).reshape(S, 1, 1)
        pred_cov_L = _chol_1x1(pred_cov, jitter)                                           # :46
        return pred_cov_L * z + pred_mean                                                  # :48,:53 (1x1 "matmul")
    Linv = ops.trtri(f).mT.contiguous()                                                    # L^-1 (lower), rows K-contiguous
    G = ops.gemm_nt(K_te_tr, Linv, uplo_b=1)                                               # K_te,tr L^-T   [S,T,N]
    pred_cov = K_te - ops.gemm_nt(G, G).to(dt)
    fc, _ = _safe_factor(pred_cov, jitter)
    return ops.gemm_nt(fc.L, z.mT.contiguous(), uplo_a=1).to(dt) + pred_mean               # :48,:53


def GeneratePrediction(train_x, train_y, test_x, pred_vol, model, latent_mean=None, theta=0.5, *, z=None):
    """voltron/rollout_utils.py:6-53.  Reads model.{log_vol_path, train_x, train_y, mean_module,
    covar_module}; returns [S, T] (T = test_x.shape[0])."""
    vol = model.log_vol_path.exp()
    if model.train_x.ndim != test_x.ndim:
        test_x_for_stack = test_x.unsqueeze(0).repeat(model.train_x.shape[0], 1)
    else:
        test_x_for_stack = test_x
    if vol.ndim == 1:
        vol_for_stack = vol.unsqueeze(0).repeat(pred_vol.shape[0], 1)
    else:
        vol_for_stack = vol
    full_x = torch.cat((model.train_x, test_x_for_stack), dim=-1)
    full_vol = torch.cat((vol_for_stack, pred_vol), dim=-1)
    idx_cut = model.train_x.shape[-1]

    train_mean = model.mean_module(model.train_x)
    train_diffs = model.train_y.unsqueeze(-1) - train_mean.unsqueeze(-1)
    test_mean_term = model.mean_module(test_x).detach().T.unsqueeze(-1)                    # :39
    S, T = full_vol.shape[0], test_x.shape[0]
    if z is None:
        z = torch.randn(S, T, 1).to(test_x.device)                                         # :47
    out = _posterior_draw(model.covar_module, full_x, full_vol, idx_cut, train_diffs, test_mean_term,
                          z.reshape(S, T, 1), 1e-4, latent_mean, theta)
    return out.squeeze(-1)


def _model_generate_prediction(model, test_x, pred_vol, n_sample=1):
    """The model-method twins, VoltronGP.py:62-95 / VoltMagpie.py:67-99 (default jitter, n_sample
    draws per test point, mean from model.train_inputs)."""
    if model.train_x.ndim != test_x.ndim:
        test_x_for_stack = test_x.unsqueeze(0).repeat(model.train_x.shape[0], 1)
    else:
        test_x_for_stack = test_x
    full_x = torch.cat((model.train_x, test_x_for_stack), dim=-1)
    full_vol = torch.cat((model.log_vol_path.exp(), pred_vol), dim=-1)
    idx_cut = model.train_x.shape[-1]
    train_mean = model.mean_module(*model.train_inputs).detach()
    train_diffs = model.train_y.unsqueeze(-1) - train_mean.reshape(model.train_y.shape).unsqueeze(-1)
    test_mean_term = model.mean_module(test_x).detach().unsqueeze(-1)
    batch = full_vol.shape[:-1]
    T = test_x.shape[0]
    z = torch.randn(*batch, T, n_sample).to(test_x.device)
    S = 1 if len(batch) == 0 else batch[0]
    out = _posterior_draw(model.covar_module, full_x, full_vol, idx_cut, train_diffs, test_mean_term,
                          z.reshape(S, T, n_sample), None)
    out = out.reshape(*batch, T, n_sample)
    return out.squeeze(-1)                                  # (samples + pred_mean).squeeze(-1), VoltMagpie.py:96-99: [T] for n_sample = 1


def rollout_engine_max_h():
    from .rollout_engine import MAX_H
    return MAX_H


def Rollouts(train_x, train_y, test_x, model, nsample=50, method="volt", theta=None, *, pred_vol=None, z=None,
             engine=None):
    """voltron/rollout_utils.py:57-93.  train_x [N], train_y [N+1] raw prices, test_x [H] ->
    samples [nsample, H] on the CPU (log-price units).  Mutates ``model`` like the reference."""
    if method != "volt":
        return nonvol_rollouts(train_x, train_y, test_x, model, nsample=nsample)
    engine = engine or os.environ.get("VOLT_ROLLOUT_ENGINE", "bordered")
    if theta is None:
        latent_mean = None
    else:
        latent_mean = train_y.log().mean()
    ntest = test_x.numel()
    if pred_vol is None:
        pred_vol = model.vol_model(test_x).sample(torch.Size((nsample,))).exp()           # :66
    pred_vol = pred_vol.to(train_x.device)
    if z is not None:
        z = z.to(train_x.device)
    if engine == "bordered" and ntest > rollout_engine_max_h():
        engine = "dense"                         # the bordered kernel holds <= 1024 appended points per sample
    if engine == "bordered":
        from .rollout_engine import rollouts_bordered
        out = rollouts_bordered(train_x, train_y, test_x, model, pred_vol, z, latent_mean, theta)
        if out is not None:
            return out
        engine = "dense"                         # a mean module the bordered kernel has no mode for: the general route
    if engine != "dense":
        raise ValueError(f"unknown rollout engine {engine!r}")

    samples = torch.zeros(nsample, ntest)                                                  # on the CPU, :65
    th = 0.5 if theta is None else theta
    samples[:, 0] = GeneratePrediction(train_x, train_y, test_x[0].unsqueeze(0), pred_vol[:, 0].unsqueeze(1),
                                       model, latent_mean, th, z=None if z is None else z[:, 0:1]).squeeze().cpu()
    train_stack_y = train_y[1:].log().repeat(nsample, 1)
    train_stack_vol = model.log_vol_path.repeat(nsample, 1)

    for idx in range(1, ntest):
        stack_y = torch.cat((train_stack_y, samples[:, :idx].to(train_stack_y.device)), -1)
        stack_vol = torch.cat((train_stack_vol, pred_vol[:, :idx].to(train_stack_vol.device).log()), -1)
        rolling_x = torch.cat((train_x, test_x[:idx]))
        model.mean_module.train_y = stack_y
        model.mean_module.train_x = rolling_x
        model.train_x = rolling_x
        model.train_y = stack_y
        model.log_vol_path = stack_vol
        samples[:, idx] = GeneratePrediction(train_x, train_y, test_x[idx].unsqueeze(0),
                                             pred_vol[:, idx].unsqueeze(-1), model, latent_mean, th,
                                             z=None if z is None else z[:, idx:idx + 1]).squeeze().cpu()
    return samples


def nonvol_rollouts(train_x, train_y, test_x, model, nsample=50, *, z=None):
    """voltron/rollout_utils.py:95-115 -- SURVEY 8(f) row 2: sequential rollouts of a baseline GP (generic kernel,
    usually a moving-average mean).  The reference conditions ``nsample`` stacked series on one more point per step
    through botorch's ``model.posterior``; here one shared factorisation + one GEMM + the mean recursion
    (rollout_engine.rollouts_shared) produce the same draws.  The model is left in the state the reference leaves it
    in (:106-111); samples come back on the CPU, log-price units.  ``z`` [nsample, H] optionally fixes the N(0,1) draws."""
    from . import rollout_engine
    ntest = test_x.numel()
    samples = rollout_engine.rollouts_shared(train_x, train_y, test_x, model, nsample, z=z)
    if ntest > 1:
        stack_y = torch.cat((train_y.log().repeat(nsample, 1), samples[:, :ntest - 1]), -1)
        rolling_x = torch.cat((train_x, test_x[:ntest - 1]))
        model.mean_module.train_y = stack_y
        model.mean_module.train_x = rolling_x
        model.train_inputs = (rolling_x.view(-1, 1),)
        model.train_targets = stack_y
        model.train()
    return samples.cpu()
