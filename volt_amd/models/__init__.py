from .BMGP import BMGP                                   # voltron/models/__init__.py:1-6 (hot-path subset)
from .VoltronGP import VoltronGP
from .single_task_variational_gp import SingleTaskVariationalGP
from .VoltMagpie import VoltMagpie
