#!/usr/bin/env python3
"""Launcher form of the alias hook:  python <repo>/compat/run_reference.py slam.py config/run_ncd128.yaml
(run from the reference's root) == `python3 slam.py config/run_ncd128.yaml` with the four hot-path modules answered by
the product (compat/clid_alias.py).  slam.py itself is not modified."""
import os
import runpy
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import clid_alias  # noqa: E402

if __name__ == "__main__":
    if len(sys.argv) < 2:
        raise SystemExit("usage: run_reference.py <script.py> [args...]")
    clid_alias.install()
    script = sys.argv[1]
    sys.argv = sys.argv[1:]
    sys.path[0] = os.path.dirname(os.path.abspath(script))  # what `python script.py` would have put there
    runpy.run_path(script, run_name="__main__")
