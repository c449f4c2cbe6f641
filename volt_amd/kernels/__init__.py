from .BMKernel import BMKernel                       # voltron/kernels/__init__.py:1-5 (hot-path subset)
from .FBMKernel import FBMKernel
from .VolKernel import VolatilityKernel, CumTrapz
