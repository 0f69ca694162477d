"""Import alias: ``import genre_shapehd_amd`` loads the package that lives in the
directory ``genre-shapehd_amd/`` (a hyphen is not importable)."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "genre-shapehd_amd")
_spec = importlib.util.spec_from_file_location(
    "genre_shapehd_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["genre_shapehd_amd"] = _mod
_spec.loader.exec_module(_mod)
