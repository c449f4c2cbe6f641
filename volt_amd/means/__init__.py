from .loglinear_mean import LogLinearMean                                             # voltron/means/__init__.py:1-3
from .EWMA import EWMA, EWMAMean, DEWMAMean, TEWMAMean, MeanRevertingEMAMean, HEWMAMean
