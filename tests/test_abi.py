"""CPU-only: the C-ABI library loads, exports every symbol include/clid_native.h declares, its structs
have the layout the ctypes mirror assumes, and the product path fails loudly without a GPU."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest
import torch

import clid_slam_amd
from clid_slam_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "clid_native.h")


def declared_functions():
    txt = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    return sorted(set(re.findall(r"\b(clid_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names = declared_functions()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), f"{n} declared in clid_native.h but not exported"
    assert sorted(_lib.EXPORTS) == [n for n in names]  # the ctypes table covers the whole header
    assert lib.clid_abi_version() == 8


def test_struct_layout_matches_the_header(tmp_path):
    src = tmp_path / "sz.c"
    src.write_text(
        '#include <stdio.h>\n#include <stddef.h>\n#include "%s"\n'
        "int main(){printf(\"%%zu %%zu %%zu %%zu %%zu %%zu %%zu %%zu %%zu %%zu %%zu %%zu %%zu %%zu\\n\", sizeof(clid_map_view),"
        " sizeof(clid_train_args), sizeof(clid_adam_args), offsetof(clid_map_view, log2cap),"
        " offsetof(clid_train_args, grad), offsetof(clid_adam_args, n_feat), sizeof(clid_cloud_view),"
        " sizeof(clid_sampler_params), offsetof(clid_cloud_view, resolution), offsetof(clid_sampler_params, pose),"
        " sizeof(clid_track_call), offsetof(clid_track_call, pc_imu), offsetof(clid_train_args, sched), offsetof(clid_train_args, eik_inv_n));"
        " return 0;}\n" % HEADER
    )
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", str(src), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    want = [C.sizeof(_lib.MapView), C.sizeof(_lib.TrainArgs), C.sizeof(_lib.AdamArgs), _lib.MapView.log2cap.offset,
            _lib.TrainArgs.grad.offset, _lib.AdamArgs.n_feat.offset, C.sizeof(_lib.CloudView),
            C.sizeof(_lib.SamplerParams), _lib.CloudView.resolution.offset, _lib.SamplerParams.pose.offset,
            C.sizeof(_lib.TrackCall), _lib.TrackCall.pc_imu.offset, _lib.TrainArgs.sched.offset, _lib.TrainArgs.eik_inv_n.offset]
    assert got == want


def test_workspace_query_and_error_reporting():
    lib = _lib.load()
    assert lib.clid_train_workspace_floats(16384, 10, 1) > 16384
    assert lib.clid_train_workspace_floats(0, 10, 1) < 0
    rc = lib.clid_adam_step(None, None, None, None, 10, 0.01, 0.9, 0.99, 1e-15, 0.0, 1, 1, None)
    assert rc == -1 and b"clid_adam_step" in lib.clid_last_error()
    with pytest.raises(RuntimeError, match="clid_adam_step"):
        _lib.check(rc, "clid_adam_step")


@pytest.mark.parametrize("bs,offset,decim,mode", [(16384, 0, 10, 1), (4096, 0, 10, 1), (2048, 2048, 10, 1),
                                                   (8192, 8192 * 3, 10, 1), (100, 7, 10, 1), (64, 0, 1, 1),
                                                   (50, 3, 2, 1), (333, 5, 25, 1), (1000, 0, 10, 0), (7, 0, 10, 1)])
def test_fused_task_map_covers_the_batch_exactly_once(bs, offset, decim, mode):
    """Every batch position is trained once; positions on the GLOBAL decimation lattice
    ((offset + p) % decim == 0, utils/mapper.py:701-702) get exactly 6 shifted copies."""
    lib = _lib.load()
    main = np.zeros(bs, np.int32)
    fd = np.zeros(bs, np.int32)
    nt = C.c_int32(0)
    rc = lib.clid_debug_task_cover(bs, offset, decim, mode, main.ctypes.data, fd.ctypes.data, C.addressof(nt))
    assert rc == 0
    assert (main == 1).all(), np.nonzero(main != 1)[0][:10]
    lattice = ((np.arange(bs) + offset) % decim == 0) if mode == 1 else np.zeros(bs, bool)
    assert (fd[lattice] == 6).all() and (fd[~lattice] == 0).all()
    assert nt.value * 8 >= bs


def test_no_cpu_fallback():
    """The hot methods raise instead of silently computing on the CPU."""
    import shim_io

    cfg = shim_io.config(device="cpu")
    nm = shim_io.neural_points(cfg)
    dec = shim_io.decoder(cfg)
    x = torch.zeros(4, 3)
    with pytest.raises(RuntimeError, match="GPU"):
        nm.query_feature(x)
    with pytest.raises(RuntimeError, match="GPU"):
        dec.sdf(torch.zeros(4, 11))
    with pytest.raises(RuntimeError, match="GPU"):
        nm.radius_neighborhood_search(x)
    mp, _ = shim_io.mapper(cfg, nm, dec)
    with pytest.raises(RuntimeError):
        mp.mapping(1)


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "clid-slam_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in txt.replace("oracle/", "").lower() or f in ("neural_points.py",), f


def _mix64(seed, counter, e):
    M = (1 << 64) - 1
    z = (seed + 0x9E3779B97F4A7C15 * (counter + 1) + 0xD1B54A32D192ED03 * (e + 1)) & M
    for _ in range(2):
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
        z ^= z >> 31
    return z


def test_batch_draw_generator_host_restatement():
    """The counter-based generator of clid_mapping_prep (batch composition of utils/mapper.py:473-500): the host entry point
    equals the published splitmix64 arithmetic restated here, stays in range and is uniform (chi-square, 16 bins)."""
    lib = _lib.load()
    for seed, counter, e, rng in ((42, 1, 0, 10), (42, 1, 1, 897000), (2**63 + 5, 77, 123456789, 10_000_000), (0, 0, 0, 1)):
        assert lib.clid_debug_prep_draw(seed, counter, e, rng) == (_mix64(seed, counter, e) * rng) >> 64
    draws = np.array([lib.clid_debug_prep_draw(42, 3, e, 16) for e in range(32000)])
    assert draws.min() == 0 and draws.max() == 15
    counts = np.bincount(draws, minlength=16)
    chi2 = float(((counts - 2000.0) ** 2 / 2000.0).sum())
    assert chi2 < 45.0  # 15 degrees of freedom: p(chi2 > 45) < 1e-4
    # consecutive positions / consecutive calls are decorrelated
    a = np.array([lib.clid_debug_prep_draw(42, 3, e, 1 << 20) for e in range(4000)], dtype=np.float64)
    b = np.array([lib.clid_debug_prep_draw(42, 4, e, 1 << 20) for e in range(4000)], dtype=np.float64)
    assert abs(np.corrcoef(a[:-1], a[1:])[0, 1]) < 0.06 and abs(np.corrcoef(a, b)[0, 1]) < 0.06
    rc = lib.clid_mapping_prep(None, 8, None, 0, 0, 0, 0, None, 0, 0, 0, None, 0.4, None, 0, 0, 1, None)
    assert rc == -1 and b"clid_mapping_prep" in lib.clid_last_error()
