"""Stands in for the reference's cffi extension package `_ext` (built there by
torch.utils.ffi, toolbox/*/build.py): re-exports the ctypes shim object(s) of
genre-shapehd_amd/_loader.py under the reference's names.  The loader is located from this
file's own path and shared process-wide, so the reference's import lines work whether this
tree is imported as `genre_shapehd_amd.toolbox...`, as `toolbox...` or as `nndistance...`."""
import importlib.util
import os
import sys

_NAME = "_genre_shapehd_amd_loader"


def _loader():
    if _NAME in sys.modules:
        return sys.modules[_NAME]
    here = os.path.dirname(os.path.abspath(__file__))
    pkg = os.path.normpath(os.path.join(here, *([".."] * 3)))
    spec = importlib.util.spec_from_file_location(_NAME, os.path.join(pkg, "_loader.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[_NAME] = mod
    try:
        spec.loader.exec_module(mod)
    except BaseException:
        del sys.modules[_NAME]
        raise
    return mod


my_lib = _loader().my_lib
