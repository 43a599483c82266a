"""CPU: host logic of the raw-point map shim (clid_slam_amd.LocalPointCloudMap: insert / window / table rebuild)
against the reference-generated fixture G9.  The kernels behind region_specific_sdf_estimation / DataSampler are
covered by the GPU tests (tests/test_sampler_gpu.py)."""
import numpy as np
import pytest
import torch

import golden_io as gio
from clid_slam_amd import DataSampler, HotPathConfig, LocalPointCloudMap
from clid_slam_amd.tools import transform_torch


def _cfg(g, device="cpu"):
    cfg = HotPathConfig()
    cfg.device = device
    cfg.local_buffer_size = int(gio.S(g["local_buffer_size"]))
    cfg.local_map_size = float(gio.S(g["local_map_size"]))
    return cfg


def test_raw_point_map_maintenance_matches_g9():
    g = gio.load("g9_sampler.npz")
    lpm = LocalPointCloudMap(_cfg(g))
    assert lpm.neighbor_idx.shape == (7, 3) and abs(lpm.max_valid_range - 0.6928) < 1e-6
    for fid in range(int(gio.S(g["n_frames"]))):
        pts, pose = gio.T(g[f"f{fid}_points"]), gio.T(g[f"f{fid}_pose"])
        lpm.update_map(pose[:3, 3], transform_torch(pts, pose))
        assert np.abs(lpm.local_point_cloud_map.numpy() - g[f"f{fid}_cloud"]).max() <= 2e-6  # transform rounding only
        occ = torch.nonzero(lpm.buffer_pt_index >= 0).flatten()
        assert np.array_equal(occ.numpy(), g[f"f{fid}_slot"])
        assert np.array_equal(lpm.buffer_pt_index[occ].numpy(), g[f"f{fid}_slot_idx"])


def test_no_cpu_fallback_for_the_sampler():
    g = gio.load("g9_sampler.npz")
    cfg = _cfg(g)
    lpm = LocalPointCloudMap(cfg)
    pts, pose = gio.T(g["f0_points"]), gio.T(g["f0_pose"])
    lpm.update_map(pose[:3, 3], transform_torch(pts, pose))
    with pytest.raises(RuntimeError, match="GPU"):
        lpm.region_specific_sdf_estimation(transform_torch(pts, pose))
    with pytest.raises(RuntimeError, match="GPU"):
        DataSampler(cfg).sample(pts, lpm, pose)
