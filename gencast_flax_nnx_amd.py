"""Importable alias for the package directory `gencast-flax-nnx_amd/`.

A hyphen cannot appear in a Python module name, so `import gencast_flax_nnx_amd`
loads the package that lives in `gencast-flax-nnx_amd/` (same object, sub-modules
resolve inside that directory).
"""
import importlib.util as _ilu
import os as _os
import sys as _sys

_dir = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "gencast-flax-nnx_amd")
_spec = _ilu.spec_from_file_location(
    __name__, _os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = _ilu.module_from_spec(_spec)
_sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
