"""Imported by Python's `site` at start-up when this directory is on PYTHONPATH: installs the alias finder of
compat/clid_alias.py, then runs the sitecustomize this one shadows (if the installation has one)."""
import importlib.util
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, _HERE) if _HERE not in sys.path else None
import clid_alias  # noqa: E402

clid_alias.install()
for _p in sys.path:  # chain to the installation's own sitecustomize, e.g. /usr/lib/python3/sitecustomize.py
    _f = os.path.join(_p or ".", "sitecustomize.py")
    if os.path.isfile(_f) and os.path.abspath(_f) != os.path.abspath(__file__):
        _spec = importlib.util.spec_from_file_location("_shadowed_sitecustomize", _f)
        try:
            _spec.loader.exec_module(importlib.util.module_from_spec(_spec))
        except Exception:  # pragma: no cover - never let the host's hook break start-up
            pass
        break
