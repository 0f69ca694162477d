"""`from nndistance.functions import nndistance, nndistance_w_idx, nndistance_score` as in the reference
(functions/__init__.py:1)"""
from . import nnd as _m

__all__ = ["nndistance", "nndistance_w_idx", "nndistance_score"]
globals().update({name: getattr(_m, name) for name in __all__})
