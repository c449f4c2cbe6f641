from .BMGP import BMGP                                   # voltron/models/__init__.py:1-6 (hot-path subset)
from .VoltronGP import VoltronGP
from .VoltMagpie import VoltMagpie
