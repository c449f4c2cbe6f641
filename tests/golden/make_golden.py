#!/usr/bin/env python3
"""Generate golden input/output vectors by EXECUTING the reference's own hot-path files.

Runs only in the build container (needs /root/reference); the GPU box never sees the
reference, only the ``*.npz`` fixtures this writes next to itself.

How the reference is loaded: ``import voltron`` fails because gpytorch and
botorch are not installed, so the three hot-path files are loaded one at a time with
``importlib`` after ``sys.modules`` is given stand-ins for exactly the gpytorch names they
import at module level:

    gpytorch.kernels.Kernel          -> empty nn.Module base whose __call__ wraps forward()
                                        in an object with .evaluate()  (lazy-tensor stand-in)
    gpytorch.means.Mean              -> empty nn.Module base, __call__ = forward
    gpytorch.utils.cholesky.psd_safe_cholesky -> torch.linalg.cholesky_ex + the published
                                        jitter loop (a RESTATEMENT, not gpytorch code)

Everything else that runs -- CumTrapz, VolatilityKernel.forward, EWMA and the four mean
classes, GeneratePrediction, Rollouts -- is the reference's code under real torch (CPU).
Rollouts' ``model.vol_model(test_x).sample(...)`` (out of scope) is a fake
that returns a fixed log-vol sample; ``torch.randn`` draws are recorded by wrapping the function during the call.

Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py
"""
import importlib.util
import os
import sys
import types
import warnings

sys.dont_write_bytecode = True          # never write __pycache__ into /root/reference
import numpy as np
import torch

REF = "/root/reference/voltron/"
OUT = os.path.dirname(os.path.abspath(__file__))
warnings.filterwarnings("ignore", message="torch.meshgrid")


def _install_standins():
    gp = types.ModuleType("gpytorch")
    k = types.ModuleType("gpytorch.kernels")
    m = types.ModuleType("gpytorch.means")
    u = types.ModuleType("gpytorch.utils")
    c = types.ModuleType("gpytorch.utils.cholesky")

    class _Lazy:
        def __init__(self, t):
            self.t = t

        def evaluate(self):
            return self.t

    class Kernel(torch.nn.Module):
        def __init__(self, **kw):
            super().__init__()

        def __call__(self, *a, **kw):
            return _Lazy(self.forward(*a, **kw))

    class Mean(torch.nn.Module):
        def __call__(self, *a, **kw):
            return self.forward(*a, **kw)

    def psd_safe_cholesky(A, upper=False, out=None, jitter=None, max_tries=3):
        L, info = torch.linalg.cholesky_ex(A)
        if not torch.any(info):
            return L
        if jitter is None:
            jitter = 1e-6 if A.dtype == torch.float32 else 1e-8
        Aprime = A.clone()
        jitter_prev = 0
        for i in range(max_tries):
            jitter_new = jitter * (10 ** i)
            Aprime.diagonal(dim1=-2, dim2=-1).add_(jitter_new - jitter_prev)
            jitter_prev = jitter_new
            L, info = torch.linalg.cholesky_ex(Aprime)
            if not torch.any(info):
                return L
        raise RuntimeError("not positive definite")

    k.Kernel, m.Mean, c.psd_safe_cholesky = Kernel, Mean, psd_safe_cholesky
    gp.kernels, gp.means, gp.utils, u.cholesky = k, m, u, c
    for name, mod in (("gpytorch", gp), ("gpytorch.kernels", k), ("gpytorch.means", m),
                      ("gpytorch.utils", u), ("gpytorch.utils.cholesky", c)):
        sys.modules[name] = mod


def _load(name, rel):
    spec = importlib.util.spec_from_file_location(name, REF + rel)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


class _RecordRandn:
    """Record every torch.randn draw the reference makes (rollout_utils.py:47) so the oracle
    and the HIP path can consume the identical N(0,1) numbers."""

    def __enter__(self):
        self.draws = []
        self._orig = torch.randn

        def rec(*a, **k):
            r = self._orig(*a, **k)
            self.draws.append(r.clone())
            return r
        torch.randn = rec
        return self

    def __exit__(self, *exc):
        torch.randn = self._orig


def sde_series(n, seed, dt=1.0 / 252):
    """example.ipynb cells 2-3 (Euler CEV/SABR-like SDE), steps = n + 1, vol-of-vol scaled so
    alpha^2 T keeps the notebook's value.  Same recipe as volt_amd.synthetic.sde_series."""
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


def main():
    _install_standins()
    VK = _load("ref_volkernel", "kernels/VolKernel.py")
    EW = _load("ref_ewma", "means/EWMA.py")
    RU = _load("ref_rollout", "rollout_utils.py")
    g = torch.Generator().manual_seed(2019)

    # ---- sanity: the torch-CPU cumsum semantics the oracle mimics (double accumulate) ----
    t = torch.rand(3000, generator=g) * 1e-3
    assert np.array_equal(torch.cumsum(t, -1).numpy(),
                          np.cumsum(t.numpy().astype(np.float64)).astype(np.float32))

    # ---- a1/a2: CumTrapz + VolatilityKernel.forward ------------------------------------
    fill = {}
    kern = VK.VolatilityKernel()
    for tag, (b, n, dt) in {"n7": (None, 7, 1 / 252), "n64": (None, 64, 1 / 252),
                            "n257": (None, 257, 1 / 365), "b3n50": (3, 50, 1 / 252),
                            "b2n130": (2, 130, 1 / 365)}.items():
        x = torch.arange(n) * dt
        shape = (n,) if b is None else (b, n)
        vol = torch.rand(*shape, generator=g) * 0.3 + 0.1
        fill[f"{tag}_x"] = x.numpy()
        fill[f"{tag}_vol"] = vol.numpy()
        fill[f"{tag}_V"] = VK.CumTrapz(vol * vol, x).numpy()
        xx = x if b is None else x.unsqueeze(0).repeat(b, 1)     # models pass [T,N,1] (VoltMagpie.py:46)
        fill[f"{tag}_K"] = kern.forward(xx.unsqueeze(-1), vol.unsqueeze(-1)).numpy()
        fill[f"{tag}_diag"] = kern.forward(xx.unsqueeze(-1), vol.unsqueeze(-1), diag=True).numpy()
    # fp64 inputs keep fp64 (dtype is inherited)
    x = (torch.arange(33) / 252.).double()
    vol = (torch.rand(33, generator=g) * 0.3 + 0.1).double()
    fill["f64_x"], fill["f64_vol"] = x.numpy(), vol.numpy()
    fill["f64_K"] = kern.forward(x.unsqueeze(-1), vol.unsqueeze(-1)).numpy()
    np.savez_compressed(os.path.join(OUT, "fill.npz"), **fill)

    # ---- a4: EWMA family ---------------------------------------------------------------
    ew = {}
    for tag, (b, n, k) in {"n60k5": (None, 60, 5), "n60k25": (None, 60, 25),
                           "b3n40k7": (3, 40, 7), "n300k100": (None, 300, 100)}.items():
        shape = (n,) if b is None else (b, n)
        y = (torch.randn(*shape, generator=g) * 0.02).cumsum(-1) + 2.3
        x = torch.arange(n) / 252.
        ew[f"{tag}_y"], ew[f"{tag}_x"] = y.numpy(), x.numpy()
        ew[f"{tag}_ewma"] = EW.EWMA(y, k).numpy()
        for cname, cls in (("ewma", EW.EWMAMean), ("dewma", EW.DEWMAMean), ("tewma", EW.TEWMAMean),
                           ("meanrevert", EW.MeanRevertingEMAMean)):
            mod = cls(x, y, k)
            ew[f"{tag}_{cname}_train"] = mod.forward(x).numpy()            # x == train_x branch
            ew[f"{tag}_{cname}_one"] = mod.forward(x[-1:] + 1 / 252.).numpy()   # numel == 1 branch
            ew[f"{tag}_{cname}_other"] = mod.forward(x[: n // 2]).numpy()  # fallthrough branch
    np.savez_compressed(os.path.join(OUT, "ewma.npz"), **ew)

    # ---- a7/a8: GeneratePrediction + Rollouts ------------------------------------------
    class FakeVolModel:
        """Stands in for the BM-GP vol model (out of scope): returns a fixed log-vol draw."""

        def __init__(self, logv):
            self.logv = logv

        def __call__(self, test_x):
            return self

        def sample(self, size):
            return self.logv

    class FakeModel:
        pass

    ro = {}
    for tag, (n, S, H, k, mean_cls, theta) in {
            "ewma": (48, 4, 6, 5, EW.EWMAMean, None),
            "ewma_theta": (48, 4, 5, 5, EW.EWMAMean, 0.05),
            "dewma": (40, 3, 4, 7, EW.DEWMAMean, None),
            "tewma": (40, 3, 4, 7, EW.TEWMAMean, None),
            "ewma_n120": (120, 5, 8, 25, EW.EWMAMean, None)}.items():
        F, vol = sde_series(n, 2019 + len(ro))
        dt = 1 / 252.
        train_y = torch.tensor(F)                         # N+1 prices
        train_x = torch.arange(n) * dt                    # N points (GenerateMultiMeanPreds.py:89)
        test_x = torch.arange(H) * dt + train_x[-1] + train_x[1]            # :90
        vol_path = torch.tensor(vol)
        logv = (torch.randn(S, H, generator=g) * 0.05).cumsum(-1) + vol_path[-1].log()
        model = FakeModel()
        model.train_x = train_x
        model.train_y = train_y[1:].log()                 # VoltMagpie(train_x, train_y.log()) w/ train_y[1:]
        model.log_vol_path = vol_path.log()
        model.mean_module = mean_cls(train_x, train_y[1:].log(), k)
        model.covar_module = VK.VolatilityKernel()
        model.vol_model = FakeVolModel(logv)
        torch.manual_seed(77)
        with _RecordRandn() as rec:       # EWMA() builds a Conv1d per call, whose init also draws
            samples = RU.Rollouts(train_x, train_y, test_x, model, nsample=S, theta=theta)
        zs = torch.stack(rec.draws, 1)[:, :, 0, 0]                                 # [S,H]
        ro[f"{tag}_train_x"], ro[f"{tag}_train_y"] = train_x.numpy(), train_y.numpy()
        ro[f"{tag}_test_x"], ro[f"{tag}_vol_path"] = test_x.numpy(), vol_path.numpy()
        ro[f"{tag}_pred_vol"], ro[f"{tag}_z"] = logv.exp().numpy(), zs.numpy()
        ro[f"{tag}_k"] = np.array(k)
        ro[f"{tag}_theta"] = np.array(np.nan if theta is None else theta)
        ro[f"{tag}_samples"] = samples.numpy()

    # multi-point GeneratePrediction with a non-EWMA mean (GenerateMultiMeanPreds.py:115-119)
    n, S, T = 36, 3, 5
    F, vol = sde_series(n, 4242)
    train_y = torch.tensor(F)
    train_x = torch.arange(n) / 252.
    test_x = torch.arange(T) / 252. + train_x[-1] + train_x[1]
    pred_vol = (torch.randn(S, T, generator=g) * 0.05).cumsum(-1).exp() * vol[-1]
    const = 2.25

    class ConstMean(torch.nn.Module):
        def forward(self, x):
            return torch.full((x.shape[0],), const)

    model = FakeModel()
    model.train_x, model.train_y = train_x, train_y[1:].log()
    model.log_vol_path = torch.tensor(vol).log()
    model.mean_module = ConstMean()
    model.covar_module = VK.VolatilityKernel()
    torch.manual_seed(5)
    with _RecordRandn() as rec:
        gp_out = RU.GeneratePrediction(train_x, train_y, test_x, pred_vol, model)
    z = rec.draws[0]
    ro.update(gpm_train_x=train_x.numpy(), gpm_train_y=train_y.numpy(), gpm_test_x=test_x.numpy(),
              gpm_vol_path=vol, gpm_pred_vol=pred_vol.numpy(), gpm_z=z.numpy(),
              gpm_const=np.array(const), gpm_samples=gp_out.numpy())
    np.savez_compressed(os.path.join(OUT, "rollouts.npz"), **ro)

    # ---- (f)2: nonvol_rollouts (rollout_utils.py:95-115) -------------------------------------------------
    # The loop is the reference's; ``model.posterior`` (botorch, absent) is a dense fp64 exact-GP predictive written
    # out here, fed by the reference's own EWMA mean classes.  Pins: what is stacked, which mean branch is hit,
    # how the draws are consumed.
    def matern25(a, b, ls, os_):
        d = (a.reshape(-1, 1) - b.reshape(1, -1)).abs() / ls
        return os_ * (1 + np.sqrt(5) * d + 5. / 3. * d ** 2) * torch.exp(-np.sqrt(5) * d)

    def rbf(a, b, ls, os_):
        d = (a.reshape(-1, 1) - b.reshape(1, -1)) / ls
        return os_ * torch.exp(-0.5 * d ** 2)

    class FakePosterior:
        def __init__(self, mean, cov):
            self.mean_, self.cov = mean, cov            # mean [q] or [S,q]; cov [q,q]

        def sample(self, sample_shape=torch.Size()):
            L = torch.linalg.cholesky(self.cov)
            eps = torch.randn(*sample_shape, *self.mean_.shape)
            return (self.mean_ + (L @ eps.double().unsqueeze(-1)).squeeze(-1)).float().unsqueeze(-1)

    class FakeExactGP:
        def __init__(self, train_x, log_y, mean_module, kfun, noise):
            self.train_inputs, self.train_targets = (train_x.view(-1, 1),), log_y
            self.mean_module, self.kfun, self.noise = mean_module, kfun, noise

        def train(self):
            return self

        def posterior(self, X):
            xt = self.train_inputs[0].reshape(-1)
            xs = X.reshape(-1)
            y = self.train_targets.double()
            full_mean = self.mean_module(torch.cat((xt, xs)).view(-1, 1)).double()      # third branch of EWMAMean.forward
            n = xt.numel()
            mt, ms = full_mean[..., :n], full_mean[..., n:]
            xt, xs = xt.double(), xs.double()
            Ktt = self.kfun(xt, xt) + self.noise * torch.eye(n, dtype=torch.float64)
            Kst = self.kfun(xs, xt)
            sol = torch.linalg.solve(Ktt, (y - mt).unsqueeze(-1)).squeeze(-1)
            mean = ms + sol @ Kst.T
            cov = self.kfun(xs, xs) - Kst @ torch.linalg.solve(Ktt, Kst.T)
            return FakePosterior(mean, cov)

    nv = {}
    for tag, (n, S, H, k, mean_cls, kf, ls, os_, noise) in {
            "matern_ewma": (60, 4, 6, 5, EW.EWMAMean, matern25, 0.3, 0.5, 0.05),
            "rbf_dewma": (50, 3, 5, 7, EW.DEWMAMean, rbf, 0.2, 0.3, 0.1),
            "matern_tewma": (80, 5, 8, 10, EW.TEWMAMean, matern25, 0.5, 0.2, 0.02)}.items():
        F, _ = sde_series(n - 1, 3000 + len(nv))
        train_y = torch.tensor(F)                                       # n prices; nonvol uses all of them (:100)
        train_x = torch.arange(n) / 252.
        test_x = torch.arange(H) / 252. + train_x[-1] + train_x[1]
        model = FakeExactGP(train_x, train_y.log(), mean_cls(train_x, train_y.log(), k),
                            lambda a, b, kf=kf, ls=ls, os_=os_: kf(a, b, ls, os_), noise)
        torch.manual_seed(99)
        with _RecordRandn() as rec:
            samples = RU.Rollouts(train_x, train_y, test_x, model, nsample=S, method="nonvol")
        zs = torch.stack([d.reshape(-1) for d in rec.draws], 1)                         # [S,H]
        nv.update({f"{tag}_train_x": train_x.numpy(), f"{tag}_train_y": train_y.numpy(), f"{tag}_test_x": test_x.numpy(),
                   f"{tag}_z": zs.numpy(), f"{tag}_k": np.array(k), f"{tag}_ls": np.array(ls), f"{tag}_os": np.array(os_),
                   f"{tag}_noise": np.array(noise), f"{tag}_samples": samples.numpy()})
    np.savez_compressed(os.path.join(OUT, "nonvol.npz"), **nv)

    for f in ("fill.npz", "ewma.npz", "rollouts.npz", "nonvol.npz"):
        print(f, os.path.getsize(os.path.join(OUT, f)), "bytes")


if __name__ == "__main__":
    main()
