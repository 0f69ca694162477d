"""`from toolbox.calc_prob.calc_prob.functions import CalcStopProb` works as in the reference (functions/__init__.py:1)"""
from . import calc_prob as _m

CalcStopProb = _m.CalcStopProb
__all__ = ["CalcStopProb"]
