"""Import hook that lets the reference's `slam.py` run unmodified on the product classes.

`slam.py:23-30` (and `utils/mesher.py:13-14`, `utils/error_state_iekf.py:9-10`) import the hot-path classes by module path:
`model.decoder.Decoder`, `model.neural_points.NeuralPoints`, `model.local_point_cloud_map.LocalPointCloudMap`,
`utils.mapper.Mapper`.  A script's own directory precedes PYTHONPATH on `sys.path`, so a shadowing `model/` package
could never win against the reference's; instead a meta-path finder answers exactly those four module names with the
product modules' contents.  Every other `model.*` / `utils.*` module still loads from the reference's files.
Installed by `compat/sitecustomize.py` (PYTHONPATH=<repo>/compat) or by `compat/run_reference.py` (a launcher)."""
import importlib
import importlib.abc
import importlib.util
import os
import sys

ALIASES = {
    "model.decoder": "clid_slam_amd.decoder",                              # model/decoder.py:12
    "model.neural_points": "clid_slam_amd.neural_points",                  # model/neural_points.py:25
    "model.local_point_cloud_map": "clid_slam_amd.local_point_cloud_map",  # model/local_point_cloud_map.py:11
    "utils.mapper": "clid_slam_amd.mapper",                                # utils/mapper.py:35
}
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _AliasFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if fullname in ALIASES:
            return importlib.util.spec_from_loader(fullname, self, origin=f"alias of {ALIASES[fullname]}")
        return None

    def create_module(self, spec):
        return None  # a fresh module object; exec_module fills it

    def exec_module(self, module):
        if _ROOT not in sys.path:
            sys.path.append(_ROOT)  # `clid_slam_amd` (alias module at the repository root)
        src = importlib.import_module(ALIASES[module.__spec__.name])
        module.__dict__.update((k, v) for k, v in vars(src).items() if not k.startswith("__"))
        module.__doc__ = f"alias of {src.__name__} (compat/clid_alias.py)"


def install() -> None:
    if not any(isinstance(f, _AliasFinder) for f in sys.meta_path):
        sys.meta_path.insert(0, _AliasFinder())
