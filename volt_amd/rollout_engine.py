"""Host side of the bordered rollout engine (csrc/rollout.hip, DESIGN.md "Rollouts").

Per series, once:  U = CumTrapz weights with the LAST train weight un-halved (the halving moves to
the test point, VolKernel.py:8-9 applied to the stacked path of rollout_utils.py:17-20), and the two
scalars the shared train block K_NN = fill(U) contributes to every sample's bordered system,
rho = u'K_NN^-1 u and tau = u'K_NN^-1 r_tr (train_block_terms).  Then ONE kernel launch walks all H
horizon steps for every sample.
"""
from __future__ import annotations

import warnings

import torch

from . import _lib, ops
from .gp import NotPSDError, NumericalWarning, _safe_factor, _dense as _dense_cov
from .means import DEWMAMean, EWMAMean, MeanRevertingEMAMean, TEWMAMean

_MODES = {EWMAMean: 0, DEWMAMean: 1, TEWMAMean: 2, MeanRevertingEMAMean: 3}


def _tail(series: torch.Tensor, k: int) -> torch.Tensor:
    """Last k entries of `series` left-padded with its first value (the conv padding of EWMA.py:27-28)."""
    n = series.shape[-1]
    if n >= k:
        return series[..., n - k:].contiguous()
    pad = series[..., :1].expand(*series.shape[:-1], k - n)
    return torch.cat((pad, series), -1).contiguous()


def _family_state(log_y, k, mean_mode, mr_theta=0.5, mr_latent=None):
    """Train-point means of the EWMA family for log_y [G,N] and the k-long tails the rollout kernels continue from:
    (m_tr [G,N], hist_y, hist_e1, hist_e2 [G,k], ema_prev [G], mr_latent [G], taps [k])."""
    f32 = torch.float32
    G, N = log_y.shape
    w = ops.ewma_weights(k, log_y.device)
    ema = ops.ewma(log_y, k)                                                   # [G,N+1]
    hist_e1 = hist_e2 = ema_prev = mrl = None
    if mean_mode == 0:
        m_tr = ema[:, :-1]
    elif mean_mode == 1:
        ee = ops.ewma(ema, k)[:, :-1]                                          # [G,N+1]
        m_tr = (2 * ema - ee)[:, :-1]
        hist_e1, hist_e2 = _tail(ema[:, :-1], k), None
    elif mean_mode == 2:
        ee = ops.ewma(ema, k)[:, :-1]
        eee = ops.ewma(ee, k)[:, :-1]
        m_tr = (3 * ema - 3 * ee + eee)[:, :-1]
        hist_e1, hist_e2 = _tail(ema[:, :-1], k), _tail(ee[:, :-1], k)
    elif mean_mode == 3:
        mrl = (log_y.mean(-1) if mr_latent is None else mr_latent.reshape(-1).expand(G)).to(f32).contiguous()
        em = ema.clone()
        em[:, 1:] -= mr_theta * (ema[:, :-1] - mrl[:, None])
        m_tr = em[:, :-1]
        ema_prev = ema[:, N - 1].contiguous()
    else:
        raise ValueError("mean_mode")
    return m_tr, _tail(log_y.to(f32), k), hist_e1, hist_e2, ema_prev, mrl, w


MAX_H = 1024         # VOLT_ROLLOUT_MAX_H: a lane owns 4 entries of each of up to four 256-entry chunks of a factor row


def train_block_terms(U, r_tr, solve="factor", jitter=1e-4):
    """rho = u'K^-1 u and tau = u'K^-1 r_tr for the shared train block K = fill(U), in fp64 [G].

    u, the covariance between the train points and ANY appended point, is U itself -- k(x*, x_i) = V[min(i, *)] =
    U[i] -- which is also the last column of K.  Hence K^-1 u = e_{N-1} exactly and
        rho = U[N-1],   tau = r_tr[N-1]                                     (solve="closed").
    solve="factor" -- THE DEFAULT since round 5 -- takes the general route the reference takes (rollout_utils.py:35-36:
    factor the train block, two solves) -- in fp64 on volt_potrf_f64 / volt_trsv_*_f64, because the noise-free block has
    condition number 1e6 (N = 400) .. 1e8 (N = 4096) and an fp32 factor cannot carry it.  The closed form is one of the
    min-structure identities SURVEY 4 reserves for testing: it stays available (and the tests hold the two against each
    other), but the product path no longer leans on it."""
    U64 = U.double()
    if solve == "closed":
        return U64[:, -1].contiguous(), r_tr[:, -1].double().contiguous()
    if solve != "factor":
        raise ValueError(f"unknown train-block solve {solve!r}")
    fct, _ = _safe_factor(ops.fill(U64), jitter)                 # psd_safe_cholesky(K_tr, 1e-4), :35, in fp64
    q = ops.trsv(fct, U64, check=True)                           # L^-1 u
    ztr = ops.trsv(fct, r_tr.double(), check=True)               # L^-1 r_tr
    return (q * q).sum(-1).contiguous(), (q * ztr).sum(-1).contiguous()


MODE_GIVEN = 4       # a mean that depends on x alone (constant / linear / log-linear): its values are handed in


def rollout_series(train_x, log_y, log_vol_path, test_x, pred_vol, z, mean_mode, k, latent_mean=None, theta=None,
                   mr_theta=0.5, mr_latent=None, jitter=1e-4, solve="factor", timing=None, resubstitute=False,
                   given_mean=None):
    """Batched engine entry.  train_x [N]; log_y, log_vol_path [G,N]; test_x [H]; pred_vol, z [G,S,H].
    Returns (samples [G,S,H] on the device, info [G,S]).  `timing` (a dict) receives events bracketing the kernel.
    ``mean_mode=MODE_GIVEN`` with ``given_mean=(m_train [G,N], m_test [G,H])``: a mean module that is a function of x
    alone (the weather driver's default constant mean, experiments/weather/GPGenerator.py:68-82; the stocks driver's
    constant / loglinear choices, GenerateMultiMeanPreds.py:168-177) -- the mean of an appended point is then history-free.
    ``resubstitute``: re-solve every sample's triangular system from its stored rows at every step (H^3/6 * 4 B of HBM
    traffic per path) instead of extending it by one entry -- bitwise the same paths; the cross-check of the default."""
    dev = train_x.device
    G, N = log_y.shape
    S, H = pred_vol.shape[-2:]
    if H > MAX_H:
        raise ValueError(f"bordered rollouts support horizons up to {MAX_H} steps (got {H}); use engine='dense'")
    f32 = torch.float32
    vol = log_vol_path.exp().to(f32)
    x = train_x.to(f32)
    dx = (x[1] - x[0]).reshape(1).expand(G).contiguous()
    # U: CumTrapz over [train, first test point] keeps full weight on train point N-1
    xe = torch.cat((x, test_x[:1].to(f32)))
    U = ops.cumtrapz(torch.cat((vol, vol[:, -1:]), -1), xe, square=True)[:, :N].contiguous()
    # The running sum the appended points continue from is the fp32 entry the train block itself holds for point
    # N-1 (K's entries ARE these fp32 prefixes); from there on the kernel adds the increments in fp64.
    acc0 = U[:, -1].double().contiguous()
    # train residuals with the model's mean family
    if mean_mode == MODE_GIVEN:
        m_tr, m_te = given_mean
        m_tr = m_tr.to(f32).expand(G, N)
        k = 1
        hist_y, hist_e1, hist_e2, ema_prev, mrl = _tail(log_y.to(f32), 1), m_te.to(f32).expand(G, H).contiguous(), None, None, None
        w = torch.ones(1, dtype=f32, device=dev)
    else:
        m_tr, hist_y, hist_e1, hist_e2, ema_prev, mrl, w = _family_state(log_y, k, mean_mode, mr_theta, mr_latent)
    r_tr = (log_y.to(f32) - m_tr).contiguous()
    # rho = u'K^-1 u and tau = u'K^-1 r_tr enter every sample's Schur complement C_s - rho 11', whose entries are
    # ~dx vol^2 while rho ~ V[N-1]: fp64, and the kernel keeps (CumTrapz sum - rho) in fp64 too.
    rho, tau = train_block_terms(U, r_tr, solve, jitter)
    samples = torch.empty(G, S, H, dtype=f32, device=dev)
    info = torch.empty(G, S, dtype=torch.int32, device=dev)
    scratch = None
    if resubstitute:
        nbytes = _lib.lib().volt_rollout_scratch_bytes(G, S, H)
        scratch = torch.empty(nbytes // 4, dtype=f32, device=dev)
    lat = None
    if theta is not None:
        lat = torch.as_tensor(latent_mean, dtype=f32, device=dev).reshape(-1).expand(G).contiguous()
    pv = pred_vol.to(f32).contiguous()
    zz = z.to(f32).contiguous()

    def P(t):
        return None if t is None else t.data_ptr()
    if timing is not None:
        timing["start"] = torch.cuda.Event(enable_timing=True)
        timing["stop"] = torch.cuda.Event(enable_timing=True)
        timing["start"].record()
    _lib.check(_lib.lib().volt_rollout_bordered_f32(
        P(rho), P(tau), P(acc0), P(dx), P(hist_y), P(hist_e1), P(hist_e2), P(ema_prev), P(mrl), P(lat), P(w),
        P(pv), P(zz), P(samples), P(scratch), P(info), G, S, H, k, mean_mode, int(theta is not None),
        float(theta or 0.0), float(mr_theta), float(jitter), _lib.stream_ptr()), "volt_rollout_bordered")
    if timing is not None:
        timing["stop"].record()
    return samples, info


def _depends_on_x_alone(mm):
    """Constant / linear / log-linear means (gp.ConstantMean, gp.LinearMean, means.LogLinearMean): parameters and the
    input, no series state."""
    from .gp import ConstantMean, LinearMean
    return isinstance(mm, (ConstantMean, LinearMean))


def rollouts_bordered(train_x, train_y, test_x, model, pred_vol, z, latent_mean, theta):
    """Rollouts(..., engine="bordered"): same inputs / outputs / model mutation as rollout_utils.Rollouts.  Returns None
    for a mean module this engine has no mode for (neither the EWMA family nor a function of x alone): the caller then
    runs the dense engine, which takes any mean module -- as it does for horizons beyond MAX_H."""
    mm = model.mean_module
    given = None
    if type(mm) not in _MODES:
        if not _depends_on_x_alone(mm):
            return None                                  # the caller falls back to the dense engine
        with torch.no_grad():                            # rollout_utils.py:31,39: mean_module(train_x), mean_module(test_x)
            given = (mm(train_x).reshape(1, -1), mm(test_x).reshape(1, -1))
    S, H = pred_vol.shape
    if z is None:
        z = torch.randn(S, H, device=train_x.device)
    if model.log_vol_path.ndim != 1:
        raise NotImplementedError("Rollouts expects an unbatched model (rollout_utils.py:71-72 repeats its state)")
    log_y = train_y[1:].log()
    kw = {}
    if isinstance(mm, MeanRevertingEMAMean):
        kw = dict(mr_theta=mm.theta, mr_latent=mm.latent_mean)
    if given is not None:
        kw = dict(given_mean=given)
    samples, info = rollout_series(train_x, log_y.unsqueeze(0), model.log_vol_path.unsqueeze(0), test_x,
                                   pred_vol.unsqueeze(0), z.unsqueeze(0), MODE_GIVEN if given is not None else _MODES[type(mm)],
                                   1 if given is not None else mm.k, latent_mean, theta, **kw)
    if bool((info[0] != 0).any()):
        exhausted = info[0] < 0
        if bool(exhausted.any()):           # psd_safe_cholesky(pred_cov, jitter=1e-4) raises here, rollout_utils.py:46
            raise NotPSDError(f"rollouts: predictive variance not positive after the jitter ladder for "
                              f"{int(exhausted.sum())} of {S} sample paths (first at horizon step "
                              f"{int((-info[0][exhausted]).min())})")
        bad = info[0] > 0
        warnings.warn(f"rollouts: {int(bad.sum())} of {S} sample paths hit a non-positive pivot "
                      f"(first at horizon step {int(info[0][bad].min())}); jitter was applied", NumericalWarning)
    samples = samples[0]
    # leave the model in the state the reference leaves it in (rollout_utils.py:80-86, last idx = H-1)
    if H > 1:
        stack_y = torch.cat((log_y.repeat(S, 1), samples[:, :H - 1]), -1)
        stack_vol = torch.cat((model.log_vol_path.repeat(S, 1), pred_vol[:, :H - 1].log()), -1)
        rolling_x = torch.cat((train_x, test_x[:H - 1]))
        mm.train_y, mm.train_x = stack_y, rolling_x      # (rollout_utils.py:81-82 sets these on ANY mean module)
        model.train_x, model.train_y, model.log_vol_path = rolling_x, stack_y, stack_vol
    return samples.cpu()


def rollouts_shared(train_x, train_y, test_x, model, nsample, z=None):
    """nonvol_rollouts (voltron/rollout_utils.py:95-115) for an exact GP whose kernel does not depend on the sample.

    The reference re-conditions S stacked series on one more point per step through ``model.posterior``.  All S
    systems share their matrix, and that matrix is a leading block of K([train, test]) + s2 I, so ONE factorisation
    L of the (N+H)^2 matrix serves every step: with l = L[N+i, :N+i], d = L[N+i, N+i] the latent conditional of
    point N+i is  mean = m_s(x_i) + l w_s,  var = d^2 - s2,  and the entry of w_s = L^-1 r_s the draw appends is
    sqrt(var) z / d -- independent of the history.  Hence  sample_s = m_s + c + M z_s  with c = L[N:, :N] L_NN^-1 r_tr
    and a fixed lower-triangular M: one GEMM, plus the moving-average recursion of the mean (csrc/rollout.hip).
    Returns samples [S,H] (device) and the dense pieces for the caller's bookkeeping."""
    dev = train_x.device
    f32 = torch.float32
    N, H, S = train_x.numel(), test_x.numel(), nsample
    log_y = train_y.log().to(f32).reshape(1, N)
    mm = model.mean_module
    xs = torch.cat((train_x, test_x)).to(f32).unsqueeze(-1)
    with torch.no_grad():
        K = _dense_cov(model.covar_module(xs, xs)).to(f32)
        noise = float(model.likelihood.noise.reshape(-1)[0])
        fct, _ = _safe_factor((K + noise * torch.eye(N + H, device=dev)).unsqueeze(0))
        L = fct.L[0]
        family = type(mm) in _MODES
        if family:
            kw = dict(mr_theta=mm.theta, mr_latent=mm.latent_mean) if isinstance(mm, MeanRevertingEMAMean) else {}
            m_tr, hist_y, hist_e1, hist_e2, ema_prev, mrl, w = _family_state(log_y, mm.k, _MODES[type(mm)], **kw)
        else:
            m_all = mm(xs).to(f32).reshape(-1)
            m_tr = m_all[:N].reshape(1, N)
        r = torch.zeros(1, N + H, device=dev, dtype=f32)
        r[:, :N] = log_y - m_tr
        w0 = ops.trsv(fct, r)[:, :N]                                             # L_NN^-1 r_tr (leading block of L)
        c = ops.gemm_nt(L[N:, :N].contiguous(), w0)[:, 0]                        # [H]
        d = L.diagonal()[N:]
        var = d * d - noise
        for jit in (1e-6, 1e-5, 1e-4):                                           # psd_safe_cholesky of the 1x1 covariance
            var = torch.where(var > 0, var, var + jit)
        var = var.clamp_min(0.0)
        sd = var.sqrt()
        M = torch.tril(L[N:, N:], -1) * (sd / d).unsqueeze(0) + torch.diag(sd)
        if z is None:
            z = torch.randn(S, H, device=dev)
        e = ops.gemm_nt(z.to(f32).contiguous(), M.contiguous(), uplo_b=1) + c    # [S,H]
        if not family:
            return m_all[N:].unsqueeze(0) + e
        samples = torch.empty(1, S, H, dtype=f32, device=dev)
        e3 = e.reshape(1, S, H).contiguous()

        def P(t):
            return None if t is None else t.data_ptr()
        _lib.check(_lib.lib().volt_rollout_shared_f32(
            P(hist_y), P(hist_e1), P(hist_e2), P(ema_prev), P(mrl), P(w), P(e3), P(samples), 1, S, H, mm.k,
            _MODES[type(mm)], float(kw.get("mr_theta", 0.5)), _lib.stream_ptr()), "volt_rollout_shared")
        return samples[0]
