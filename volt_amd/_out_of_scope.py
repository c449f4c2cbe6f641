"""Names the reference's drivers import unconditionally but that are OUTSIDE this package's scope (SURVEY 2 rows 7, 10,
13: the Matern / RBF / spectral-mixture baseline GPs, their kernels and TrainBasicModel -- experiments/stocks/
GenerateMultiMeanPreds.py:16,18, experiments/weather/GPGenerator.py:15, BasicWind.py).  They must RESOLVE, or
``from voltron.train_utils import ..., TrainBasicModel`` fails at import time even for ``--kernel volt`` runs.

Resolution is lazy (module ``__getattr__``, PEP 562): the name comes from ``baselines/`` of the source tree when that is
importable (the tests' baseline GPs, computing on libvolt_hip.so like everything else), otherwise it is a stand-in that
raises NotImplementedError when it is CALLED, never when it is imported."""
import importlib
import os

WHERE = {
    "MaternGP": "models", "SMGP": "models",
    "TrainBasicModel": "train",
    "ScaleKernel": "gpkernels", "RBFKernel": "gpkernels", "MaternKernel": "gpkernels", "SpectralMixtureKernel": "gpkernels",
}


def _stand_in(name):
    def _raise(*args, **kwargs):
        raise NotImplementedError(
            f"{name} is a baseline outside volt_amd's scope (the exact-GP volatility path, SURVEY.md 8); the source tree "
            f"keeps one in baselines/{WHERE[name]}.py -- put the repository root on sys.path to use it")
    if name == "TrainBasicModel":
        _raise.__name__ = name
        return _raise
    return type(name, (), {"__init__": _raise, "__doc__": f"stand-in for the out-of-scope baseline {name}"})


def resolve(name):
    if name not in WHERE:
        raise AttributeError(name)
    modname = "baselines." + WHERE[name]
    try:
        mod = importlib.import_module(modname)
    except ModuleNotFoundError as e:
        if e.name not in ("baselines", modname):     # a dependency of OUR baselines is missing: that is an error, not "out of scope"
            raise
        return _stand_in(name)
    # only the source tree's own baselines/ (next to volt_amd/) counts: an unrelated installed package of that name does not
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if os.path.dirname(os.path.dirname(os.path.abspath(getattr(mod, "__file__", "") or ""))) != here:
        return _stand_in(name)
    return getattr(mod, name)
