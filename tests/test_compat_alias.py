"""CPU: the import hook of compat/ (north_star: "slam.py is unchanged outside this path").  With PYTHONPATH=<repo>/compat
(sitecustomize) or through compat/run_reference.py, the module paths slam.py:23-30 imports resolve to the product classes
while every other `model.*` / `utils.*` module still comes from the reference tree -- whose own directory precedes
PYTHONPATH on sys.path, as it does for `python3 slam.py`.  A miniature stand-in tree is built in a temp directory
(/root/reference does not exist on the GPU box); when the real reference is present it is checked too."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PROBE = textwrap.dedent("""
    import sys
    from model.decoder import Decoder
    from model.local_point_cloud_map import LocalPointCloudMap
    from model.neural_points import NeuralPoints
    from utils.mapper import Mapper
    import clid_slam_amd
    assert Decoder is clid_slam_amd.Decoder and NeuralPoints is clid_slam_amd.NeuralPoints
    assert Mapper is clid_slam_amd.Mapper and LocalPointCloudMap is clid_slam_amd.LocalPointCloudMap
    import utils.config, model
    print("CONFIG_FROM", utils.config.__file__)
""")


def _probe(ref_root, launcher=False):
    script = os.path.join(ref_root, "probe_like_slam.py") if os.access(ref_root, os.W_OK) else None
    env = dict(os.environ)
    if script is None:  # read-only tree: the probe runs from stdin with the tree as working directory ('' on sys.path)
        cmd = [sys.executable, "-"]
    else:
        with open(script, "w") as fh:
            fh.write(PROBE)
        cmd = [sys.executable, script]
    if launcher and script is not None:
        cmd = [sys.executable, os.path.join(ROOT, "compat", "run_reference.py"), script]
    else:
        env["PYTHONPATH"] = os.path.join(ROOT, "compat")  # the ONE entry
    return subprocess.run(cmd, cwd=ref_root, env=env, capture_output=True, text=True, timeout=300,
                          input=PROBE if script is None else None)


def test_alias_packages_shadow_only_the_hot_path_modules(tmp_path):
    for pkg, mods in (("model", ("decoder", "neural_points", "local_point_cloud_map")), ("utils", ("mapper", "config", "tools"))):
        d = tmp_path / pkg
        d.mkdir()
        (d / "__init__.py").write_text("")
        for m in mods:
            (d / f"{m}.py").write_text(f"ORIGIN = 'stand-in reference {pkg}.{m}'\n")
    for launcher in (False, True):
        out = _probe(str(tmp_path), launcher)
        assert out.returncode == 0, out.stderr[-2000:]
        assert f"CONFIG_FROM {tmp_path}/utils/config.py" in out.stdout


def test_alias_packages_in_front_of_the_real_reference():
    ref = "/root/reference"
    if not os.path.isdir(os.path.join(ref, "utils")):
        import pytest

        pytest.skip("the reference tree is not present on this box")
    out = _probe(ref)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "CONFIG_FROM /root/reference/utils/config.py" in out.stdout
