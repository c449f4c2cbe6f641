"""Minimal stand-ins for the four gpytorch types the hot path touches, backed by the HIP library.

The reference's call sites are unchanged:

    likelihood = GaussianLikelihood()                               # train_utils.py:100,195
    mll = ExactMarginalLogLikelihood(likelihood, model)             # train_utils.py:127,240
    output = model(train_x)                                         # ExactGP.__call__ -> forward
    loss = -mll(output, train_y); loss.backward()                   # train_utils.py:249-250

gpytorch itself is third-party and absent from /root/reference; what is restated here (from its
published behaviour, gpytorch >= 1.0.1 per setup.py:19) is only the plumbing: parameter names and
registration order (the reference freezes parameters *positionally*, train_utils.py:226-227), the
``softplus + 1e-4`` noise constraint, input unsqueezing in ``ExactGP.__call__`` and the
``psd_safe_cholesky`` jitter policy.  The arithmetic -- K + sigma^2 I, Cholesky, solves, log-det,
gradient -- runs in libvolt_hip.so through one ``torch.autograd.Function``; there is no CPU path.
"""
from __future__ import annotations

import math
import threading
import warnings

import torch
import torch.nn.functional as F
from torch import nn

from . import ops


def same_values(a: torch.Tensor, b: torch.Tensor) -> bool:
    """``torch.equal(a, b)`` without its device synchronisation when the answer is known from the host side: two views
    of the same memory with the same layout hold the same values.  The reference's loops pass the stored training
    inputs back into ``model(train_x)`` on every iteration (train_utils.py:246), where each ``torch.equal`` on GPU
    tensors (VoltMagpie.py:122, EWMA.py:49) costs a full host-device round trip."""
    if (a.shape == b.shape and a.dtype == b.dtype and a.device == b.device and a.stride() == b.stride()
            and a.data_ptr() == b.data_ptr()):
        return True
    return a.shape == b.shape and torch.equal(a, b)


class EqualMemo:
    """Remembers that one particular tensor OBJECT compared equal to a stored tensor; valid while both version counters
    (bumped by any in-place write through any view) are unchanged.  Keyed on identity through a weak reference, never
    on addresses, so a new tensor that happens to reuse freed memory is always compared for real."""

    def __init__(self):
        self._ref, self._ver = None, None

    def equal(self, x: torch.Tensor, stored: torch.Tensor, compare) -> bool:
        held = self._ref() if self._ref is not None else None
        if held is x and self._ver == (x._version, stored._version):
            return True
        ok = bool(compare())
        if ok:
            import weakref
            self._ref, self._ver = weakref.ref(x), (x._version, stored._version)
        return ok


class NotPSDError(RuntimeError):
    pass


class deferred_checks:
    """Context in which the factorisation's per-matrix ``info`` is NOT read back on the host after every step (that
    read is a device synchronisation, and it makes a training iteration impossible to capture in a hipGraph) but
    OR-ed into a device-side accumulator; ``any_bad()`` / ``raise_if_bad()`` read it when the caller chooses.  The
    training loops of train_utils use it (graph-captured loops, and the batched eager loops, which check every few
    iterations and replay from a snapshot with gpytorch's jitter ladder when a step failed).

    ``immediate = True`` keeps the per-step host check (and the ladder) while still sizing the accumulators: one
    accumulator per (length, device) of ``info``, so an iteration that factors batches of different shapes (an MLL
    step and a GPCV step, two models) keeps every flag, and nothing is allocated -- or re-zeroed on replay -- inside
    a graph capture: run one iteration with ``immediate = True`` under the context first."""
    _active = None

    def __init__(self, immediate=False):
        self.immediate = immediate
        self._acc = {}

    def __enter__(self):
        self._prev = deferred_checks._active
        deferred_checks._active = self
        return self

    def __exit__(self, *exc):
        deferred_checks._active = self._prev
        return False

    @classmethod
    def deferring(cls):
        """The active context if it is deferring the check, else None."""
        a = cls._active
        return a if a is not None and not a.immediate else None

    def _slot(self, info):
        flat = info.reshape(-1)
        key = (flat.shape[0], flat.device)
        acc = self._acc.get(key)
        if acc is None:
            if flat.is_cuda and torch.cuda.is_current_stream_capturing():
                raise RuntimeError("deferred_checks: no accumulator for this info shape yet -- run one iteration with "
                                   "immediate=True under the context before capturing")
            acc = self._acc[key] = torch.zeros_like(flat)
        return acc, flat

    def reserve(self, info):
        self._slot(info)

    def note(self, info):
        acc, flat = self._slot(info)
        acc.bitwise_or_(flat)                            # one launch per step; non-zero once any step failed

    def clear(self):
        for acc in self._acc.values():
            acc.zero_()

    def any_bad(self) -> int:
        """Number of matrices with a failed factorisation since the last clear() -- ONE host read."""
        if not self._acc:
            return 0
        return int(torch.stack([(a != 0).sum() for a in self._acc.values()]).sum().item())

    def raise_if_bad(self):
        nbad = self.any_bad()
        if nbad and any(bool((a < 0).any().item()) for a in self._acc.values()):
            # (the accumulators OR the steps' info words: a negative one holds an internal code -- INT_MIN, INT_MIN + 1 --, not a pivot)
            raise ops._lib.VoltHipError("a factorisation reported an INTERNAL error (hand-off time-out / workspace table) while "
                                        "the check was deferred; rerun without deferral: the step is then re-run on the "
                                        "launch-per-column schedule")
        if nbad:
            raise NotPSDError(f"{nbad} matrices failed a factorisation while the check was deferred (not positive "
                              "definite); rerun without deferral to get gpytorch's jitter-retry behaviour")


class NanError(RuntimeError):
    pass


class NumericalWarning(RuntimeWarning):
    pass


# ------------------------------------------------------------------------------------ modules
class Module(nn.Module):
    """gpytorch.Module stand-in: nn.Module plus ``register_prior`` (train_utils.py:158-159 puts a NormalPrior on the
    log-linear mean's slope; ExactMarginalLogLikelihood adds the log-priors / N like gpytorch does)."""

    def register_prior(self, name, prior, param_or_closure, setting_closure=None):
        if not hasattr(self, "_priors"):
            object.__setattr__(self, "_priors", {})
        if isinstance(param_or_closure, str) and not hasattr(self, param_or_closure):
            raise AttributeError(f"Unknown parameter {param_or_closure} for {type(self).__name__}")
        self._priors[name] = (prior, param_or_closure)

    def named_priors(self, prefix=""):
        for name, (prior, target) in getattr(self, "_priors", {}).items():
            closure = (lambda m, t=target: getattr(m, t)) if isinstance(target, str) else target
            yield prefix + name, self, prior, closure
        for cname, child in self.named_children():
            if isinstance(child, Module):
                yield from child.named_priors(prefix + cname + ".")


class NormalPrior:
    """gpytorch.priors.NormalPrior(loc, scale): log-density evaluated on the parameter's own device."""

    def __init__(self, loc, scale, validate_args=None, transform=None):
        self.loc, self.scale = float(loc), float(scale)

    def log_prob(self, x):
        return -0.5 * ((x - self.loc) / self.scale) ** 2 - math.log(self.scale) - 0.5 * math.log(2 * math.pi)


class Kernel(Module):
    """gpytorch.kernels.Kernel stand-in: ``kernel(x1, x2)`` calls ``forward`` and wraps the result
    so that ``.evaluate()`` (rollout_utils.py:26) works."""
    has_lengthscale = False

    def __init__(self, **kwargs):
        super().__init__()

    def __call__(self, *args, **kwargs):
        return _Evaluated(self.forward(*args, **kwargs))


class _Evaluated:
    """Already-dense "lazy tensor"."""

    def __init__(self, t):
        self.tensor = t

    def evaluate(self):
        return self.tensor

    def to_dense(self):
        return self.tensor

    def detach(self):
        return self.tensor.detach()

    @property
    def shape(self):
        return self.tensor.shape


class _ScaledDense(_Evaluated):
    """K = scale * base with a trainable scalar `scale` and a constant dense `base` (BMKernel: vol *
    min(x, x')).  Keeping the factorisation lets the MLL return d mll / d scale from the quantities
    the fused step already has (see _ExactMLL.backward) instead of a dense d mll / d K."""

    def __init__(self, scale, base):
        self.scale, self.base = scale, base

    def _product(self, scale):
        # a batched kernel carries scale [T,1] over one shared base [N,N]
        return (scale.unsqueeze(-1) if scale.ndim > 1 else scale) * self.base

    @property
    def tensor(self):
        return self._product(self.scale)

    @tensor.setter
    def tensor(self, v):
        raise AttributeError("read-only")

    def evaluate(self):
        return self._product(self.scale)

    def to_dense(self):
        return self.evaluate()

    def detach(self):
        return self._product(self.scale.detach())

    @property
    def shape(self):
        return self.base.shape


def _dense(c):
    return c.evaluate() if isinstance(c, _Evaluated) else c


class Mean(Module):
    def __call__(self, x):
        if x.ndimension() == 1:          # gpytorch.means.Mean.__call__ adds the feature dimension to 1-D inputs
            x = x.unsqueeze(1)
        res = self.forward(x)
        return res


class ConstantMean(Mean):
    def __init__(self, batch_shape=torch.Size(), **kwargs):
        super().__init__()
        self.batch_shape = batch_shape
        self.register_parameter("constant", nn.Parameter(torch.zeros(*batch_shape, 1)))

    def forward(self, x):
        if x.shape[:-2] == self.batch_shape:
            return self.constant.expand(x.shape[:-1])
        if len(self.batch_shape) and x.ndim == 2:              # batched constant over shared inputs [N,1]
            return self.constant.expand(*self.batch_shape, x.shape[-2])
        return self.constant.expand(*x.shape[:-1]) if x.ndim > 1 else self.constant.expand(x.shape)


class LinearMean(Mean):
    def __init__(self, input_size, batch_shape=torch.Size(), bias=True):
        super().__init__()
        self.register_parameter("weights", nn.Parameter(torch.randn(*batch_shape, input_size, 1)))
        if bias:
            self.register_parameter("bias", nn.Parameter(torch.randn(*batch_shape, 1)))
        else:
            self.bias = None

    def forward(self, x):
        if x.ndim == 1:
            x = x.unsqueeze(-1)
        res = (x * self.weights.squeeze(-1).unsqueeze(-2)).sum(-1)     # x w over the (tiny) feature axis: elementwise, no BLAS
        if self.bias is not None:
            res = res + self.bias
        return res


class MultivariateNormal:
    """gpytorch.distributions.MultivariateNormal stand-in (VoltMagpie.py:127)."""

    def __init__(self, mean, covariance_matrix):
        self.loc = mean
        self._covar = covariance_matrix

    @property
    def mean(self):
        return self.loc

    @property
    def covariance_matrix(self):
        return _dense(self._covar)

    @property
    def lazy_covariance_matrix(self):
        return self._covar if isinstance(self._covar, _Evaluated) else _Evaluated(self._covar)

    @property
    def variance(self):
        return torch.diagonal(self.covariance_matrix, dim1=-2, dim2=-1)

    def rsample(self, sample_shape=torch.Size(), base_samples=None):
        cov = self.covariance_matrix
        L = psd_safe_cholesky(cov)
        shape = torch.Size(sample_shape) + self.loc.shape
        if base_samples is None:
            base_samples = torch.randn(shape, dtype=self.loc.dtype, device=self.loc.device)
        # loc + L z on the library GEMM: rows of z times L' (batched covariances: the batch leads the GEMM)
        H = L.shape[-1]
        if L.ndim == 2:
            lz = ops.gemm_nt(base_samples.reshape(-1, H), L, uplo_b=1).reshape(base_samples.shape)
        else:
            nb = L.numel() // (H * H)
            zb = base_samples.reshape(-1, nb, H).transpose(0, 1).contiguous()              # [T,S,H]
            lz = ops.gemm_nt(zb, L.reshape(nb, H, H), uplo_b=1).transpose(0, 1).reshape(base_samples.shape)
        return self.loc + lz.to(self.loc.dtype)

    def sample(self, sample_shape=torch.Size(), base_samples=None):
        with torch.no_grad():
            return self.rsample(sample_shape, base_samples)


# --------------------------------------------------------------------------------- likelihood
class _NoiseCovar(Module):
    def __init__(self, batch_shape):
        super().__init__()
        self.register_parameter("raw_noise", nn.Parameter(torch.zeros(*batch_shape, 1)))


class GaussianLikelihood(Module):
    """noise = softplus(raw_noise) + 1e-4  (gpytorch default GreaterThan(1e-4) constraint).
    ``likelihood.raw_noise.data = torch.tensor([1e-5])`` (train_utils.py:222) => noise ~= 0.6933."""
    NOISE_FLOOR = 1e-4

    def __init__(self, batch_shape=torch.Size(), **kwargs):
        super().__init__()
        self.noise_covar = _NoiseCovar(batch_shape)

    @property
    def raw_noise(self):
        return self.noise_covar.raw_noise

    @raw_noise.setter
    def raw_noise(self, value):
        self.noise_covar.raw_noise = value

    @property
    def noise(self):
        return F.softplus(self.noise_covar.raw_noise) + self.NOISE_FLOOR

    @noise.setter
    def noise(self, value):
        value = torch.as_tensor(value, dtype=self.raw_noise.dtype, device=self.raw_noise.device)
        v = (value - self.NOISE_FLOOR).clamp_min(1e-12).expand_as(self.raw_noise)
        with torch.no_grad():
            self.noise_covar.raw_noise.copy_(v + torch.log(-torch.expm1(-v)))    # inverse softplus

    def forward(self, function_dist):
        """p(y|f): add the noise to the diagonal (dense; used outside the fused MLL only)."""
        cov = function_dist.covariance_matrix
        n = cov.shape[-1]
        noise = self.noise.reshape(*self.noise.shape[:-1], 1, 1)
        return MultivariateNormal(function_dist.mean, cov + noise * torch.eye(n, dtype=cov.dtype, device=cov.device))



# -------------------------------------------------------------------------------------- model
class ExactGP(Module):
    """gpytorch.models.ExactGP stand-in.  Registers ``likelihood`` first (so it is parameter 0 for
    the positional grad_flags of train_utils.py:226), stores 1-D inputs as [N,1] and passes
    unsqueezed inputs to ``forward`` exactly like gpytorch does."""

    def __init__(self, train_inputs, train_targets, likelihood):
        super().__init__()
        self.likelihood = likelihood
        if train_inputs is not None and torch.is_tensor(train_inputs):
            train_inputs = (train_inputs,)
        self.train_inputs = None if train_inputs is None else tuple(
            t.unsqueeze(-1) if t.ndimension() == 1 else t for t in train_inputs)
        self.train_targets = train_targets

    def __call__(self, *args, **kwargs):
        inputs = [a.unsqueeze(-1) if torch.is_tensor(a) and a.ndimension() == 1 else a for a in args]
        if self.training:
            return self.forward(*inputs, **kwargs)
        return self.posterior_call(*inputs, **kwargs)

    def posterior_call(self, *inputs, **kwargs):
        raise NotImplementedError(
            f"{type(self).__name__}: eval-mode model(x) (generic gpytorch posterior) is outside the accelerated path; "
            "use GeneratePrediction / Rollouts (voltron/rollout_utils.py) which this package does implement")


# --------------------------------------------------------------------------------------- MLL
class refine_alpha:
    """Opt-in context (``with gp.refine_alpha(): loss = -mll(model(x), y)``): the fp32 MLL step refines alpha = K_s^-1 (y - m)
    by one step of iterative refinement against K itself (VOLT_REFINE_ALPHA) -- for models whose mean has trainable
    parameters (d mll / d mean = alpha / N; voltron/train_utils.py:213-220) and that train to the noise floor, where any
    fp32 factorisation leaves alpha at cond * eps (the reference's own path included).  Default off: the step then does
    what the reference's fp32 path does.  Costs one more pass over K and two triangular solves per step.

    REQUIREMENT: the residual r - K_s alpha is formed against whole rows of the caller's K, so BOTH triangles of every K
    that passes through an MLL inside the context must hold the symmetric matrix (everywhere else the library reads the
    lower triangle only); the kernels of this package return full symmetric matrices.  fp32 only (an fp64 step ignores
    it: its alpha is at cond * eps64 already).  The switch is per THREAD (threading.local) and re-entrant: a context
    entered on one thread does not change what another thread's steps do."""
    _tls = threading.local()

    def __init__(self, on=True):
        self.on = on

    @staticmethod
    def active():
        return bool(getattr(refine_alpha._tls, "on", False))

    def __enter__(self):
        self._prev = refine_alpha.active()
        refine_alpha._tls.on = self.on
        return self

    def __exit__(self, *exc):
        refine_alpha._tls.on = self._prev
        return False


class _ExactMLL(torch.autograd.Function):
    """mll[b] = log N(y_b; m_b, K_b + s2_b I) / N  with analytic backward:
    d/d s2 = 1/2 (a'a - tr K_s^-1)/N,  d/d m = a/N,  d/d y = -a/N,  a = K_s^-1 (y - m)."""

    @staticmethod
    def forward(ctx, K, mean, noise, target, holder, scale=None):
        B, n = mean.shape
        want_dk = bool(ctx.needs_input_grad[0])       # kernels with trainable parameters (Matern / SM / FBM baselines)
        K = K.detach()
        need_grad = want_dk or any(ctx.needs_input_grad[1:4]) or (scale is not None and ctx.needs_input_grad[5])
        if K.dtype != torch.float64:                  # the caller's dtype is kept (VolKernel.py:28-33): fp32 or fp64 step
            K = K.to(torch.float32)
        dt = K.dtype
        if want_dk and dt != torch.float32:
            raise NotImplementedError("the dense d mll / d K path (FBM kernel) is fp32 only")
        ws = holder.workspace(B, n, need_grad, K.device, dt)
        resid = (target - mean).to(dt)
        noise = noise.to(dt)
        # gpytorch factors through psd_safe_cholesky: plain first, then jitter 1e-6 * 10^i (fp32 default) with a
        # NumericalWarning, then NotPSDError.  Same ladder here; the jitter is added inside the fused step.
        out, alpha, info = ops.mll_step(K, resid, noise, ws, want_grad=need_grad, jitter=0.0, refine_alpha=refine_alpha.active())
        chk = deferred_checks.deferring()
        if chk is not None:
            chk.note(info)
            bad = 0
        else:
            bad = int((info != 0).sum().item())
            if deferred_checks._active is not None:
                deferred_checks._active.reserve(info)
        if bad and ops.info_internal(info):
            # NOT "not positive definite": a one-launch step's hand-off timed out or its workspace table is not what the init
            # wrote (include/volt_hip.h: info <= INT_MIN + 1).  gpytorch's ladder is for pivots; this is re-run ONCE on the
            # launch-per-column schedule (fp32: the table-free path of the same entry point) and reported.
            codes = sorted({int(v) & 0xffffffff for v in info[info <= ops._lib.INFO_INTERNAL_MAX].tolist()})
            if dt != torch.float32:
                raise ops._lib.VoltHipError(f"volt_mll_step_f64: internal error, info = {[hex(c) for c in codes]}")
            warnings.warn(f"volt_mll_step_f32 reported an internal error (info = {[hex(c) for c in codes]}) on its one-launch "
                          "schedule; the step was re-run on the launch-per-column schedule", ops._lib.VoltHipWarning)
            out, alpha, info = ops.mll_step(K, resid, noise, ws, want_grad=need_grad, jitter=0.0,
                                            refine_alpha=refine_alpha.active(), tables=False)
            if ops.info_internal(info):
                raise ops._lib.VoltHipError(f"volt_mll_step_f32: internal error on both schedules, info = {info.tolist()}")
            bad = int((info != 0).sum().item())
        if bad:
            if torch.isnan(K).any() or torch.isnan(resid).any() or torch.isnan(noise).any():
                raise NanError("cholesky: NaN in the covariance, the noise or the residual")
            first = int(info[info != 0][0].item())
            for i in range(3):
                jitter = (1e-6 if dt == torch.float32 else 1e-8) * (10 ** i)      # gpytorch's defaults per dtype
                out, alpha, info = ops.mll_step(K, resid, noise, ws, want_grad=need_grad, jitter=jitter, refine_alpha=refine_alpha.active())
                if ops.info_internal(info):
                    raise ops._lib.VoltHipError(f"volt_mll_step: internal error, info = {info.tolist()}")
                if not bool((info != 0).any().item()):
                    warnings.warn(f"A not p.d., added jitter of {jitter:.1e} to the diagonal", NumericalWarning)
                    break
            else:
                raise NotPSDError(f"K + sigma^2 I not positive definite for {bad} of {B} series after jitter "
                                  f"(first failing pivot {first})")
        ctx.n = n
        ctx.has_scale = scale is not None
        ctx.want_dk = want_dk
        if need_grad:
            # the step's outputs live in the workspace, which the next forward overwrites: ONE packed copy of what backward
            # needs ([B, 8 + N]: the scalars and alpha) instead of a clone per tensor -- the iteration of a short series is
            # launch-bound (profiles/r06/pipeline_kernel_stats.csv: ~45 tiny kernels beside a 0.13 ms step), every copy is one
            pk = torch.cat((out, alpha), dim=1)
            keep = lambda t: t.detach().clone() if t.is_leaf else t.detach()     # (computed tensors are fresh already)
            extra = (pk[:, 2:6], keep(noise), keep(scale)) if scale is not None else ()
            gk = (ops.mll_grad_k(ws),) if want_dk else ()       # 1/2 (a a' - K_s^-1) / N from the step's own Y = L^-T
            ctx.save_for_backward(pk[:, 1], pk[:, 8:], *gk, *extra)
            return pk[:, 0]
        return out[:, 0].clone()

    @staticmethod
    def backward(ctx, g):
        dsig, alpha = ctx.saved_tensors[:2]
        gm = g.unsqueeze(-1) * alpha / ctx.n
        gscale = gK = None
        rest = ctx.saved_tensors[2:]
        if ctx.want_dk:
            gK, rest = g.reshape(-1, 1, 1) * rest[0], rest[1:]
        if ctx.has_scale:
            # K = c M  =>  a'Ma = (r'a - s2 a'a)/c  and  tr(K_s^-1 M) = (N - s2 tr K_s^-1)/c
            q, noise, c = rest
            quad, tr, aa = q[:, 0], q[:, 2], q[:, 3]
            gscale = (g * 0.5 * ((quad - noise * aa) - (ctx.n - noise * tr)) / (ctx.n * c.reshape(-1))).reshape(c.shape)
            if gscale.shape != c.shape:
                gscale = gscale.sum().reshape(c.shape)
        return gK, gm, g * dsig, -gm, None, gscale


class ExactMarginalLogLikelihood(Module):
    """gpytorch.mlls.ExactMarginalLogLikelihood stand-in: ``mll(model(x), y)`` returns the marginal
    log likelihood divided by the number of data points (a scalar, or [T] for a batched model)."""

    def __init__(self, likelihood, model):
        super().__init__()
        # plain attributes: gpytorch registers both as submodules, but nobody iterates mll.parameters()
        object.__setattr__(self, "likelihood", likelihood)
        object.__setattr__(self, "model", model)
        self._ws = None

    def workspace(self, B, n, want_grad, device, dtype=torch.float32):
        ws = self._ws
        if ws is None or not ws.fits(B, n, want_grad, dtype) or ws.buf.device != device:
            self._ws = ws = ops.MllWorkspace(B, n, want_grad, device, dtype)
        return ws

    def forward(self, function_dist, target):
        mean = function_dist.mean
        lazy = function_dist.lazy_covariance_matrix
        scale = None
        if isinstance(lazy, _ScaledDense):
            scale = lazy.scale
            K = lazy._product(lazy.scale.detach())
        else:
            K = function_dist.covariance_matrix
        batched = mean.ndim > 1
        n = mean.shape[-1]
        mean2, K3, t2 = mean.reshape(-1, n), K.reshape(-1, n, n), target.reshape(-1, n)
        if not K3.is_cuda:
            raise ops._lib.VoltHipError("ExactMarginalLogLikelihood: tensors must live on the MI355X; no CPU fallback")
        noise = self.likelihood.noise.reshape(-1)
        noise = noise.expand(mean2.shape[0]) if noise.numel() == 1 else noise
        dt = torch.float64 if K3.dtype == torch.float64 else torch.float32      # computed in the covariance's dtype
        res = _ExactMLL.apply(K3, mean2.to(dt), noise.to(dt), t2.to(dt), self, scale)
        res = res.reshape(mean.shape[:-1]) if batched else res.reshape(())
        priors = self.model.named_priors() if isinstance(self.model, Module) else ()
        for _, module, prior, closure in priors:                          # gpytorch: + sum log p(theta) / num_data
            res = res + prior.log_prob(closure(module)).sum() / n
        return res


# -------------------------------------------------------------------------- psd_safe_cholesky
def psd_safe_cholesky(A, upper=False, out=None, jitter=None, max_tries=3):
    """gpytorch.utils.cholesky.psd_safe_cholesky on the HIP potrf: try once; on a non-positive
    pivot add jitter * 10^i (i = 0..max_tries-1; default 1e-6 fp32 / 1e-8 fp64) to the diagonal and
    retry, warning like gpytorch; raise NanError / NotPSDError otherwise.  Returns dense L."""
    f, _ = _safe_factor(A, jitter, max_tries)
    L = f.L.reshape(A.shape)
    return L.transpose(-1, -2) if upper else L


def _safe_factor(A, jitter=None, max_tries=3):
    """psd_safe_cholesky's retry policy around the HIP potrf, in A's own dtype (fp32 or fp64; anything else is
    computed in fp32)."""
    n = A.shape[-1]
    A3 = A.reshape(-1, n, n)
    if A3.dtype != torch.float64:
        A3 = A3.to(torch.float32)
    f = ops.potrf(A3)
    if not bool((f.info != 0).any().item()):
        return f, 0.0
    tables = True
    if ops.info_internal(f.info):
        # an internal error of the one-launch factorisation (time-out / workspace table), not a pivot: once more on the
        # launch-per-column schedule, and that schedule for the rest of this call
        warnings.warn(f"volt_potrf reported an internal error (info = {f.info.tolist()[:4]} ...) on its one-launch schedule; "
                      "re-run on the launch-per-column schedule", ops._lib.VoltHipWarning)
        tables = False
        f = ops.potrf(A3, tables=False)
        if ops.info_internal(f.info):
            raise ops._lib.VoltHipError(f"volt_potrf: internal error on both schedules, info = {f.info.tolist()[:8]}")
        if not bool((f.info != 0).any().item()):
            return f, 0.0
    if torch.isnan(A3).any():
        raise NanError(f"cholesky_cpu: {int(torch.isnan(A3).sum())} of {A3.numel()} elements of the {tuple(A.shape)} tensor are NaN.")
    if jitter is None:
        jitter = 1e-6 if A.dtype == torch.float32 else 1e-8
    for i in range(max_tries):
        jitter_new = jitter * (10 ** i)
        f = ops.potrf(A3, jitter=jitter_new, tables=tables)
        if ops.info_internal(f.info):
            raise ops._lib.VoltHipError(f"volt_potrf: internal error, info = {f.info.tolist()[:8]}")
        if not bool((f.info != 0).any().item()):
            warnings.warn(f"A not p.d., added jitter of {jitter_new:.1e} to the diagonal", NumericalWarning)
            return f, jitter_new
    raise NotPSDError(f"Matrix not positive definite after repeatedly adding jitter up to {jitter_new:.1e}.")


LOG_2PI = math.log(2 * math.pi)
