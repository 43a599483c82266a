"""Pin the CPU oracle of the sample/label generation (oracle/sampler_ref.py, "next" row N2) against outputs
of the reference itself: fixtures G9 (LocalPointCloudMap + DataSampler) and G10 (Mapper.process_frame pool
bookkeeping), produced by oracle/make_golden.py --only-g9.  CPU only."""
import numpy as np
import pytest
import torch

import golden_io as gio
from oracle import sampler_ref as R


@pytest.fixture(scope="module")
def g9():
    return gio.load("g9_sampler.npz")


def _cloud(g):
    return R.LocalCloud.empty(resolution=0.2, buffer_size=int(gio.S(g["local_buffer_size"])),
                              map_size=float(gio.S(g["local_map_size"])))


def test_neighbourhood_of_the_raw_point_map():
    dx, rng = R.cloud_neighborhood(1, 0.2, 0.2)
    assert dx.shape == (7, 3)  # centre + 6 face neighbours
    assert abs(rng - 1.732 * 2 * 0.2) < 1e-12


def test_voxel_down_sample_matches_the_product_rule():
    # two restatements of utils/tools.py:639-682 (this one with scatter-amin, the product's with a sort)
    from clid_slam_amd.tools import voxel_down_sample_torch

    g = torch.Generator().manual_seed(3)
    pts = torch.rand((5000, 3), generator=g) * torch.tensor([8.0, 6.0, 3.0]) - 2.0
    a = R.voxel_down_sample(pts, 0.2)
    b = voxel_down_sample_torch(pts, 0.2)
    assert torch.equal(a, b)


def test_g9_local_cloud_and_sampler(g9):
    g = g9
    lc = _cloud(g)
    cfg = R.SamplerConfig()
    n_dropped = 0
    for fid in range(int(gio.S(g["n_frames"]))):
        pts, pose = gio.T(g[f"f{fid}_points"]), gio.T(g[f"f{fid}_pose"])
        R.cloud_update(lc, pose[:3, 3], R.transform(pts, pose))
        assert np.array_equal(lc.points.numpy(), g[f"f{fid}_cloud"])  # bit-exact raw-point map
        occ = torch.nonzero(lc.buffer_pt_index >= 0).flatten()
        assert np.array_equal(occ.numpy(), g[f"f{fid}_slot"])
        assert np.array_equal(lc.buffer_pt_index[occ].numpy(), g[f"f{fid}_slot_idx"])
        # direct region-specific estimate on mixed surface / free-space samples
        d, ok = R.region_sdf(lc, gio.T(g[f"f{fid}_q"]))
        assert np.array_equal(ok.numpy(), g[f"f{fid}_q_ok"])
        assert np.abs(d.numpy() - g[f"f{fid}_q_sdf"]).max() <= 1e-6
        # the sampler with the same draws
        noise = gio.sampler_noise(gio.S(g[f"f{fid}_seed"]), pts.shape[0])
        coord, label, weight = R.sample_region_specific(cfg, pts, lc, pose, noise)
        assert coord.shape[0] == g[f"f{fid}_coord"].shape[0]
        assert np.array_equal(coord.numpy(), g[f"f{fid}_coord"])
        assert np.abs(label.numpy() - g[f"f{fid}_label"]).max() <= 1e-6
        assert np.array_equal(weight.numpy(), g[f"f{fid}_weight"])
        n_dropped += pts.shape[0] * 8 - coord.shape[0]
        # the fixture exercises all three label kinds: plane fit, nearest point, dropped
        lab = np.abs(g[f"f{fid}_q_sdf"])
        assert (~g[f"f{fid}_q_ok"]).any() and g[f"f{fid}_q_ok"].any() and (lab < 0.05).any()
    assert n_dropped > 0


def test_g9_projective_sampler(g9):
    g = g9
    pts = gio.T(g["f0_points"])
    coord, label, weight = R.sample_projective(R.SamplerConfig(), pts, gio.sampler_noise(gio.S(g["pin_seed"]), pts.shape[0]))
    assert np.array_equal(coord.numpy(), g["pin_coord"])
    assert np.array_equal(label.numpy(), g["pin_label"])
    assert np.array_equal(weight.numpy(), g["pin_weight"])
    assert coord.shape[0] == pts.shape[0] * 8 and (weight < 0).sum() == pts.shape[0] * 3


def test_g10_pool_bookkeeping(g9):
    g = gio.load("g10_process_frame.npz")
    lc = R.LocalCloud.empty(resolution=0.2, buffer_size=int(gio.S(g["local_buffer_size"])),
                            map_size=float(gio.S(g["local_map_size"])))
    pool = R.PoolState.empty()
    cfg = R.SamplerConfig()
    for fid in range(3):
        pts, pose = gio.T(g9[f"f{fid}_points"]), gio.T(g9[f"f{fid}_pose"])
        R.cloud_update(lc, pose[:3, 3], R.transform(pts, pose))
        noise = gio.sampler_noise(gio.S(g[f"f{fid}_seed"]), pts.shape[0])
        coord, label, weight = R.sample_region_specific(cfg, pts, lc, pose, noise)
        R.pool_append_and_filter(pool, coord, label, weight, fid, pose, float(gio.S(g["window_radius"])), int(1e7))
        assert pool.pool_sample_count == int(gio.S(g[f"f{fid}_pool_count"]))
        assert pool.cur_sample_count == int(gio.S(g[f"f{fid}_cur_count"]))
        assert np.array_equal(torch.bincount(pool.time.long(), minlength=3).numpy(), g[f"f{fid}_time_hist"])
        assert abs(float(pool.sdf_label.double().sum()) - float(gio.S(g[f"f{fid}_label_sum"]))) <= 1e-3
    assert np.array_equal(pool.global_coord.numpy(), g["final_global_coord"])
    assert np.array_equal(pool.coord.numpy(), g["final_coord"])
    assert np.abs(pool.sdf_label.numpy() - g["final_label"]).max() <= 1e-6
    assert np.array_equal(pool.weight.numpy(), g["final_weight"])
    assert np.array_equal(pool.time.numpy(), g["final_time"])


def test_adaptive_iteration_offset_rule():
    assert R.adaptive_iter_offset(0.01, 3) == -5
    assert R.adaptive_iter_offset(0.10, 3) == 0
    assert R.adaptive_iter_offset(0.20, 3) == 5
    assert R.adaptive_iter_offset(0.35, 3) == 5      # restart needs frame_id > freeze_after_frame
    assert R.adaptive_iter_offset(0.35, 41) == 10
    assert R.adaptive_iter_offset(0.35, 41, adaptive=False) == 0
