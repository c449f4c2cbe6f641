#!/usr/bin/env python3
"""Golden vectors for the reference-owned half of the GPCV stage, made by EXECUTING the reference's
own code:

    voltron/models/single_task_variational_gp.py:204-254   SingleTaskVariationalGP.initialize_variational_parameters
    voltron/likelihoods/volatility_likelihood.py:42-50     VolatilityGaussianLikelihood.forward  ("exp" parameterisation)
    voltron/kernels/BMKernel.py:38-52                      BMKernel.forward  (the prior covariance kuu of the start-up)

Runs only in the build container (needs /root/reference); writes ``gpcv.npz`` next to itself.  The three files import
gpytorch and botorch at module level (absent here), so ``sys.modules`` gets stand-ins for exactly the names they import.
Most are empty classes (only needed for the ``import`` lines and as base classes).  The ones that take part in the
arithmetic are dense restatements of the gpytorch objects' documented meaning -- NOT gpytorch code:

    gpytorch.lazify(t) / LazyTensor   -> _Dense(t): .evaluate() .t() .matmul() .add_jitter(j) = + j I,
                                         .inv_matmul(r) = solve, .cholesky() = lower psd_safe_cholesky factor,
                                         .root_decomposition(method="cholesky").root = the same factor
    gpytorch.constraints.Interval     -> lower + (upper - lower) sigmoid(raw)  (BMKernel's vol parameter)
    Kernel.__call__(x)                -> _Dense(forward(x, x))

What executes is the reference's own statement sequence: the running std and its log, the clamped diag-embedded inverse
Hessian, S = L (L'HL + I)^-1 L', chol_variational_covar = 10 * chol(S).tril(), the mean constant; the "exp" likelihood's
scale = exp(f).clamp(min=1e-3).  The ELBO arithmetic (gpytorch's VariationalELBO / quadrature / KL) stays UNPINNED.

Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_gpcv.py
"""
import importlib.util
import os
import sys
import types
import warnings

sys.dont_write_bytecode = True
import numpy as np
import torch

REF = "/root/reference/voltron/"
OUT = os.path.dirname(os.path.abspath(__file__))
warnings.filterwarnings("ignore")


def psd_safe_cholesky(A, max_tries=3):
    """gpytorch utils/cholesky.py, restated (as in make_golden.py)."""
    L, info = torch.linalg.cholesky_ex(A)
    if not torch.any(info):
        return L
    jitter = 1e-6 if A.dtype == torch.float32 else 1e-8
    Ap, prev = A.clone(), 0.0
    for i in range(max_tries):
        new = jitter * (10 ** i)
        Ap.diagonal(dim1=-2, dim2=-1).add_(new - prev)
        prev = new
        L, info = torch.linalg.cholesky_ex(Ap)
        if not torch.any(info):
            return L
    raise RuntimeError("not positive definite")


class _Dense:
    """Dense stand-in for the gpytorch LazyTensor methods initialize_variational_parameters calls."""

    def __init__(self, t):
        self.tensor = t

    def evaluate(self):
        return self.tensor

    def t(self):
        return _Dense(self.tensor.mT)

    def matmul(self, other):
        return self.tensor @ (other.tensor if isinstance(other, _Dense) else other)     # LazyTensor @ Tensor -> Tensor

    def add_jitter(self, j=1e-3):
        n = self.tensor.shape[-1]
        return _Dense(self.tensor + j * torch.eye(n, dtype=self.tensor.dtype))

    def inv_matmul(self, rhs):
        return torch.cholesky_solve(rhs, psd_safe_cholesky(self.tensor))

    def cholesky(self, upper=False):
        return _Dense(psd_safe_cholesky(self.tensor))

    def root_decomposition(self, method=None):
        assert method == "cholesky"
        return types.SimpleNamespace(root=_Dense(psd_safe_cholesky(self.tensor)))


class _Interval:
    def __init__(self, lower, upper, **kw):
        self.lower, self.upper = float(lower), float(upper)

    def transform(self, raw):
        return self.lower + (self.upper - self.lower) * torch.sigmoid(raw)

    def inverse_transform(self, v):
        u = (v - self.lower) / (self.upper - self.lower)
        return torch.log(u) - torch.log1p(-u)


class _Positive:
    def __init__(self, *a, **kw):
        pass

    def transform(self, raw):
        return torch.nn.functional.softplus(raw)


class _Module(torch.nn.Module):
    """The slice of gpytorch.Module the loaded classes use: register_constraint / initialize."""

    def register_constraint(self, name, constraint):
        object.__setattr__(self, name + "_constraint", constraint)

    def initialize(self, **kw):
        for k, v in kw.items():
            getattr(self, k).data.copy_(torch.as_tensor(v).to(getattr(self, k)).expand_as(getattr(self, k)))
        return self


class _Kernel(_Module):
    def __init__(self, batch_shape=torch.Size(), **kw):
        super().__init__()
        self._bs = batch_shape

    @property
    def batch_shape(self):
        return self._bs

    def __call__(self, x1, x2=None, **kw):
        return _Dense(self.forward(x1, x1 if x2 is None else x2, **kw))


def _install_standins():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    empty = lambda n: type(n, (_Module,), {})
    deco = lambda *a, **k: (a[0] if a and callable(a[0]) and not k else (lambda f: f))
    mod("gpytorch", lazify=lambda t: t if isinstance(t, _Dense) else _Dense(t), Module=_Module)
    mod("gpytorch.constraints", Positive=_Positive, Interval=_Interval)
    mod("gpytorch.kernels", Kernel=_Kernel, ScaleKernel=empty("ScaleKernel"), RBFKernel=empty("RBFKernel"),
        InducingPointKernel=empty("InducingPointKernel"))
    mod("gpytorch.likelihoods", Likelihood=empty("Likelihood"), _OneDimensionalLikelihood=empty("_OneDimensionalLikelihood"),
        GaussianLikelihood=empty("GaussianLikelihood"), FixedNoiseGaussianLikelihood=empty("FixedNoiseGaussianLikelihood"))
    mod("gpytorch.likelihoods.gaussian_likelihood", _GaussianLikelihoodBase=empty("_GaussianLikelihoodBase"))
    mod("gpytorch.distributions", MultivariateNormal=empty("MultivariateNormal"))
    mod("gpytorch.lazy", CholLazyTensor=_Dense, TriangularLazyTensor=_Dense)
    mod("gpytorch.means", ConstantMean=empty("ConstantMean"), ZeroMean=empty("ZeroMean"))
    mod("gpytorch.models", ApproximateGP=empty("ApproximateGP"))
    mod("gpytorch.utils")
    mod("gpytorch.utils.errors", NotPSDError=RuntimeError)
    mod("gpytorch.utils.memoize", cached=deco, add_to_cache=deco, clear_cache_hook=deco)
    mod("gpytorch.variational", CholeskyVariationalDistribution=empty("CholeskyVariationalDistribution"),
        UnwhitenedVariationalStrategy=empty("UnwhitenedVariationalStrategy"), VariationalStrategy=empty("VariationalStrategy"))
    mod("botorch")
    mod("botorch.models", SingleTaskGP=empty("SingleTaskGP"))
    mod("botorch.models.gpytorch", GPyTorchModel=empty("GPyTorchModel"))
    mod("botorch.posteriors", GPyTorchPosterior=empty("GPyTorchPosterior"))


def _load(name, rel):
    spec = importlib.util.spec_from_file_location(name, REF + rel)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def main():
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(OUT, "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)                                          # for its sde_series (the fixtures' series recipe)
    sde_series = mg.sde_series
    _install_standins()
    BM = _load("ref_bmkernel", "kernels/BMKernel.py")
    VL = _load("ref_vol_likelihood", "likelihoods/volatility_likelihood.py")
    ST = _load("ref_stvgp", "models/single_task_variational_gp.py")
    out = {}
    lik = VL.VolatilityGaussianLikelihood(param="exp")
    g = torch.Generator().manual_seed(11)
    fs = torch.cat((torch.randn(40, generator=g) * 3.0, torch.tensor([-20.0, -6.9078, -6.9, 0.0, 5.0])))   # below / at / above the clamp
    out["lik_f"], out["lik_scale"] = fs.numpy(), lik.forward(fs).scale.numpy()
    out["lik_f64"], out["lik_scale64"] = fs.double().numpy(), lik.forward(fs.double()).scale.numpy()

    for tag, (n, seed, dt_, dtype) in {"n60_f64": (60, 2019, 1 / 252., torch.float64), "n120_f64": (120, 7, 1 / 252., torch.float64),
                                       "n90_f32": (90, 31, 1 / 252., torch.float32),
                                       "n80_wind_f64": (80, 5, 1 / 365., torch.float64)}.items():
        F, _ = sde_series(n, seed)
        train_x = (torch.arange(n + 1) * dt_).to(dtype)[:n]              # N inducing points = train inputs (train_utils.py:26-31)
        train_y = torch.tensor(F).to(dtype)                              # N+1 prices
        dt = train_x[1] - train_x[0]
        yy = (train_y[1:] - train_y[:-1]) / train_y[:-1] / dt ** 0.5     # train_utils.py:16-18 (restated: LearnGPCV cannot be imported)
        kern = BM.BMKernel().to(dtype)                                   # vol = 0.2 default (BMKernel.py:8), raw_vol through Interval(0,1)
        kuu_ref = kern.forward(train_x.view(-1, 1), train_x.view(-1, 1))
        fake = types.SimpleNamespace()
        fake.covar_module = kern
        dist = types.SimpleNamespace(variational_mean=torch.nn.Parameter(torch.zeros(n, dtype=dtype)),
                                     chol_variational_covar=torch.nn.Parameter(torch.eye(n, dtype=dtype)))
        fake.variational_strategy = types.SimpleNamespace(inducing_points=train_x.view(-1, 1), _variational_distribution=dist,
                                                          variational_params_initialized=torch.zeros(1))
        fake.mean_module = types.SimpleNamespace(constant=torch.nn.Parameter(torch.zeros(1, dtype=dtype)))
        ST.SingleTaskVariationalGP.initialize_variational_parameters(fake, lik, train_x, y=yy)
        assert float(fake.variational_strategy.variational_params_initialized) == 1.0
        out.update({f"{tag}_x": train_x.numpy(), f"{tag}_prices": train_y.numpy(), f"{tag}_y": yy.numpy(),
                    f"{tag}_vol": np.array(float(kern.vol)), f"{tag}_kuu": kuu_ref.detach().numpy(),
                    f"{tag}_mean": dist.variational_mean.data.numpy(), f"{tag}_chol": dist.chol_variational_covar.data.numpy(),
                    f"{tag}_const": fake.mean_module.constant.data.numpy()})
    np.savez_compressed(os.path.join(OUT, "gpcv.npz"), **out)
    print("gpcv.npz", os.path.getsize(os.path.join(OUT, "gpcv.npz")), "bytes;", sorted(out)[:6], "...")


if __name__ == "__main__":
    main()
