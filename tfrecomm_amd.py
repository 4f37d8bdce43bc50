"""Import alias: ``import tfrecomm_amd`` -> the package in ``tf-recomm_amd/``.

The package directory carries the reference's name (with its hyphen), which is not a
Python identifier; this module loads it under an importable name.
"""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tf-recomm_amd")
_spec = importlib.util.spec_from_file_location(
    "tfrecomm_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["tfrecomm_amd"] = _mod
_spec.loader.exec_module(_mod)
