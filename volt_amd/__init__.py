"""volt_amd -- MI355X-native implementation of Volt's exact-GP hot path (g-benton/Volt), exposing
the reference's own ``voltron`` surface for that path (voltron/__init__.py:2-11):

    from volt_amd.kernels import VolatilityKernel
    from volt_amd.models import VoltronGP, VoltMagpie, BMGP
    from volt_amd.rollout_utils import Rollouts, GeneratePrediction
    from volt_amd.train_utils import TrainVoltMagpieModel, TrainDataModel

``volt_amd.install_as_voltron()`` aliases the package as ``voltron`` so reference drivers import it
unchanged.  Arithmetic runs in volt_amd/csrc/libvolt_hip.so (include/volt_hip.h); nothing here has a
CPU fallback.
"""
__version__ = "0.1"

from .kernels import BMKernel, VolatilityKernel            # noqa: F401
from .models import BMGP, VoltronGP, VoltMagpie            # noqa: F401
from .rollout_utils import Rollouts, GeneratePrediction    # noqa: F401
from .train_utils import LearnGPCV                        # noqa: F401


def install_as_voltron(gpytorch_standin=True):
    """Register this package (and its hot-path submodules) under the name ``voltron``; with ``gpytorch_standin`` also
    offer the stand-ins of volt_amd/gp.py as ``gpytorch`` when no real gpytorch is importable, so that call sites such
    as ``gpytorch.mlls.VariationalELBO`` (example.ipynb cell 8) resolve."""
    import sys
    if gpytorch_standin:
        from . import gpytorch_compat
        gpytorch_compat.install()
    from . import kernels, likelihoods, means, models, rollout_utils, train_utils
    me = sys.modules[__name__]
    sys.modules.setdefault("voltron", me)
    for name, mod in (("kernels", kernels), ("likelihoods", likelihoods), ("means", means), ("models", models),
                      ("rollout_utils", rollout_utils), ("train_utils", train_utils)):
        sys.modules.setdefault("voltron." + name, mod)
    return me
