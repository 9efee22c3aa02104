"""Import shim: the product package lives in the directory ``ig-gcn_amd/`` (repo naming contract),
which is not a valid Python identifier.  ``import igcn_amd`` loads that directory as the package
``igcn_amd`` (sub-modules resolve inside it)."""
import importlib.util
import os
import sys

_here = os.path.dirname(os.path.abspath(__file__))
_pkg_dir = os.path.join(_here, "ig-gcn_amd")
_spec = importlib.util.spec_from_file_location(
    "igcn_amd", os.path.join(_pkg_dir, "__init__.py"), submodule_search_locations=[_pkg_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["igcn_amd"] = _mod
_spec.loader.exec_module(_mod)
