"""Import alias: the package directory is ``clid-slam_amd/`` (not a valid Python identifier),
so ``import clid_slam_amd`` lands here and is redirected to that directory as a real package.
"""
import importlib.util
import os
import sys

_PKG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "clid-slam_amd")
_spec = importlib.util.spec_from_file_location(
    "clid_slam_amd",
    os.path.join(_PKG_DIR, "__init__.py"),
    submodule_search_locations=[_PKG_DIR],
)
_mod = importlib.util.module_from_spec(_spec)
sys.modules["clid_slam_amd"] = _mod
_spec.loader.exec_module(_mod)
