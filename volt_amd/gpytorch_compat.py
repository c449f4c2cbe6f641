"""A ``gpytorch`` namespace made of this package's stand-ins, for running the reference's drivers and notebook
unchanged where gpytorch is not installed (it is not in this image) or should not be on the path.

The reference reaches gpytorch by attribute (``gpytorch.means.ConstantMean()``, ``gpytorch.mlls.VariationalELBO(...)``,
``with gpytorch.settings.num_gauss_hermite_locs(75):`` -- example.ipynb cell 8, train_utils.py:28,44,50,100,127,
experiments/weather/GPGenerator.py:62).  ``install()`` registers exactly the names those call sites use; each one is
the class documented in volt_amd/gp.py or variational.py -- nothing here computes anything."""
import contextlib
import sys
import types

from . import gp, variational


class _NoOpSetting(contextlib.ContextDecorator):
    """Settings that only steer gpytorch's choice between Cholesky and iterative solvers (``max_cholesky_size``,
    ``fast_computations`` ...): this path always factors exactly, so they are accepted and ignored."""

    def __init__(self, *args, **kwargs):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


def _baseline_kernel(name):
    """gpytorch.kernels.{ScaleKernel,RBFKernel,MaternKernel,SpectralMixtureKernel}: imported unconditionally by the
    reference's drivers (experiments/stocks/GenerateMultiMeanPreds.py:16), out of this package's scope -- resolved lazily
    (baselines/ of the source tree, or a stand-in that raises when called)."""
    from ._out_of_scope import WHERE, resolve
    if WHERE.get(name) == "gpkernels":
        return resolve(name)
    raise AttributeError(f"module 'gpytorch.kernels' (volt_amd stand-in) has no attribute {name!r}")


def build() -> types.ModuleType:
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        return m

    cholesky = mod("gpytorch.utils.cholesky", psd_safe_cholesky=gp.psd_safe_cholesky)
    errors = mod("gpytorch.utils.errors", NotPSDError=gp.NotPSDError, NanError=gp.NanError)
    warns = mod("gpytorch.utils.warnings", NumericalWarning=gp.NumericalWarning)
    utils = mod("gpytorch.utils", cholesky=cholesky, errors=errors, warnings=warns)
    subs = {
        "means": mod("gpytorch.means", Mean=gp.Mean, ConstantMean=gp.ConstantMean, LinearMean=gp.LinearMean),
        "kernels": mod("gpytorch.kernels", Kernel=gp.Kernel, __getattr__=_baseline_kernel),
        "likelihoods": mod("gpytorch.likelihoods", GaussianLikelihood=gp.GaussianLikelihood),
        "mlls": mod("gpytorch.mlls", ExactMarginalLogLikelihood=gp.ExactMarginalLogLikelihood,
                    VariationalELBO=variational.VariationalELBO),
        "distributions": mod("gpytorch.distributions", MultivariateNormal=gp.MultivariateNormal),
        "models": mod("gpytorch.models", ExactGP=gp.ExactGP),
        "priors": mod("gpytorch.priors", NormalPrior=gp.NormalPrior),
        "settings": mod("gpytorch.settings", num_gauss_hermite_locs=variational.num_gauss_hermite_locs,
                        max_cholesky_size=_NoOpSetting, fast_computations=_NoOpSetting,
                        fast_pred_var=_NoOpSetting, cholesky_jitter=_NoOpSetting),
        "utils": utils,
    }
    root = mod("gpytorch", __version__="volt_amd-standin", Module=gp.Module, **subs)
    root.__path__ = []                                  # a package, so ``import gpytorch.mlls`` resolves
    return root


def install(force: bool = False):
    """Register the namespace as ``gpytorch`` unless a real gpytorch is importable (``force`` overrides)."""
    if not force and "gpytorch" not in sys.modules:
        try:
            import importlib.util
            if importlib.util.find_spec("gpytorch") is not None:
                return None
        except (ImportError, ValueError):
            pass
    if not force and "gpytorch" in sys.modules and getattr(sys.modules["gpytorch"], "__version__", "") != "volt_amd-standin":
        return None
    root = build()
    sys.modules["gpytorch"] = root
    for name in ("means", "kernels", "likelihoods", "mlls", "distributions", "models", "priors", "settings", "utils"):
        sys.modules["gpytorch." + name] = getattr(root, name)
    for name in ("cholesky", "errors", "warnings"):
        sys.modules["gpytorch.utils." + name] = getattr(root.utils, name)
    return root
