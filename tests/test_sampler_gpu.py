"""GPU parity of the sample + label generation ("next" row N2): HIP kernels k_region_sdf / k_sample_frame and
the Mapper.process_frame glue against the reference-generated fixtures G9 / G10 and the CPU oracle
(oracle/sampler_ref.py).

Tolerances: sample coordinates and weights follow the reference's fp32 op order (<= 1 ulp-level, 2e-6 / 1e-6);
labels from the plane fit are |n.p + d| with world coordinates of tens of metres in fp32, so two correct
implementations of the 4x3 SVD differ by ~1e-5 m there: 1e-4 m is asserted, nearest-point and projective labels
to 2e-6.  A sample within rounding of a voxel boundary or of a validity threshold can legitimately fall on the
other side, and a badly conditioned plane fit moves by more than rounding: the ORACLE names those samples
(oracle.sampler_ref.region_sdf_ambiguity, <= 1 %); every sample it does not name is held to the strict bars."""
import numpy as np
import pytest
import torch

import golden_io as gio
from oracle import sampler_ref as R

pytestmark = pytest.mark.gpu


def _cfg(g, **over):
    from clid_slam_amd import HotPathConfig

    cfg = HotPathConfig()
    cfg.device = "cuda"
    cfg.local_buffer_size = int(gio.S(g["local_buffer_size"]))
    cfg.local_map_size = float(gio.S(g["local_map_size"]))
    for k, v in over.items():
        setattr(cfg, k, v)
    return cfg


def _cloud_at(g, fid, cfg):
    """Shim raw-point map holding exactly the reference's state after frame `fid`."""
    from clid_slam_amd import LocalPointCloudMap

    lpm = LocalPointCloudMap(cfg)
    tab = torch.full((cfg.local_buffer_size,), -1, dtype=torch.int64)
    tab[gio.T(g[f"f{fid}_slot"])] = gio.T(g[f"f{fid}_slot_idx"])
    lpm.buffer_pt_index = tab.cuda()
    lpm.local_point_cloud_map = gio.T(g[f"f{fid}_cloud"]).cuda()
    return lpm


def _oracle_cloud(lpm, cfg):
    lc = R.LocalCloud.empty(resolution=0.2, buffer_size=cfg.local_buffer_size, map_size=cfg.local_map_size)
    lc.buffer_pt_index = lpm.buffer_pt_index.cpu()
    lc.points = lpm.local_point_cloud_map.cpu()
    return lc


@pytest.mark.parametrize("fid", [0, 1, 2])
def test_region_sdf_kernel_g9(fid):
    """model/local_point_cloud_map.py:98-201 against the reference's own outputs.  The surface mask is exact (identical inputs
    => identical cells).  Labels: strict (1e-4 m) on every sample the oracle does not NAME as discontinuous in fp32 rounding
    (oracle.sampler_ref.region_sdf_ambiguity: a plane fit on the eta / residual threshold, a 4th / 5th neighbour tie, a
    badly conditioned normal); the named ones are few and bounded by the label range."""
    g = gio.load("g9_sampler.npz")
    cfg = _cfg(g)
    lpm = _cloud_at(g, fid, cfg)
    q = gio.T(g[f"f{fid}_q"])
    d, ok = lpm.region_specific_sdf_estimation(q.cuda())
    ok, d = ok.cpu().numpy(), d.cpu().numpy()
    assert np.array_equal(ok, g[f"f{fid}_q_ok"])  # identical inputs => identical cells and masks
    amb = R.region_sdf_ambiguity(_oracle_cloud(lpm, cfg), q)
    named = (amb["any"] & ~amb["cell"]).numpy()   # (the cells are given here: same inputs on both sides)
    err = np.abs(d - g[f"f{fid}_q_sdf"])
    assert named.mean() <= 1e-2, named.sum()
    assert err[~named].max() <= 1e-4, (int((err[~named] > 1e-4).sum()), float(err[~named].max()))
    assert err[named].max() <= 0.4 if named.any() else True   # (plane vs nearest-point label: bounded by the neighbourhood)
    assert np.median(err) <= 2e-6


@pytest.mark.parametrize("fid", [0, 2])
def test_sampler_kernel_g9(fid):
    """utils/data_sampler.py:260-402 against the reference's own (compacted) outputs, row by row: the rows are aligned through
    the dense (ray, sample) index -- the oracle's keep mask reproduces the reference's compaction (checked: same count) -- and
    compared on the intersection of the two keep masks; a row kept by one side only must be one the oracle names (a sample within
    rounding of a voxel boundary / of the neighbourhood's range)."""
    from clid_slam_amd import DataSampler

    g = gio.load("g9_sampler.npz")
    cfg = _cfg(g)
    lpm = _cloud_at(g, fid, cfg)
    pts, pose = gio.T(g[f"f{fid}_points"]), gio.T(g[f"f{fid}_pose"])
    noise = gio.sampler_noise(gio.S(g[f"f{fid}_seed"]), pts.shape[0])
    want_c, want_l, want_w = g[f"f{fid}_coord"], g[f"f{fid}_label"], g[f"f{fid}_weight"]
    coord, label, weight, keep, n_all = DataSampler(cfg)._run(pts.cuda(), lpm, pose, noise)
    coord, label, weight, k = coord.cpu().numpy(), label.cpu().numpy(), weight.cpu().numpy(), keep.bool().cpu().numpy()
    # the oracle's dense keep mask and ambiguity list (ray-major, like the kernel's dense outputs)
    lc = _oracle_cloud(lpm, cfg)
    sc = R.SamplerConfig()
    xyz, disp, ratio, depth, disp_s, n_all_o = R._ray_samples(sc, pts, noise)
    Rn = pts.shape[0]
    n_surface = Rn * (sc.surface_sample_n + 1)
    world = R.transform(xyz[Rn:n_surface], pose)
    _, ok = R.region_sdf(lc, world)
    amb = R.region_sdf_ambiguity(lc, world)
    kp = torch.ones(Rn * n_all_o, dtype=torch.bool)
    kp[Rn:n_surface] = ok
    named = torch.zeros(Rn * n_all_o, dtype=torch.bool)
    named[Rn:n_surface] = amb["any"]
    want_k, named = R._ray_major(kp, n_all_o).numpy(), R._ray_major(named, n_all_o).numpy()
    assert n_all == n_all_o and int(want_k.sum()) == want_c.shape[0]   # the oracle's mask IS the reference's compaction
    pos = np.cumsum(want_k) - 1                                          # dense index -> row of the golden arrays
    assert not ((k != want_k) & ~named).any(), int(((k != want_k) & ~named).sum())
    assert (k != want_k).mean() <= 1e-3
    both = k & want_k
    rows = pos[both]
    assert np.abs(coord[both] - want_c[rows]).max() <= 2e-6 * 60
    assert np.abs(weight[both] - want_w[rows]).max() <= 1e-6
    err = np.abs(label[both] - want_l[rows])
    strict = ~named[both]
    assert err[strict].max() <= 1e-4, (int((err[strict] > 1e-4).sum()), float(err[strict].max()))
    assert named.mean() <= 1e-2 and np.median(err) <= 2e-6


@pytest.mark.parametrize("fid", [0, 1, 2])
def test_sampler_kernel_dense_vs_oracle(fid):
    """Dense (un-compacted) outputs against the oracle with the same draws: every row comparable, flips counted."""
    from clid_slam_amd import DataSampler

    g = gio.load("g9_sampler.npz")
    cfg = _cfg(g)
    lpm = _cloud_at(g, fid, cfg)
    pts, pose = gio.T(g[f"f{fid}_points"]), gio.T(g[f"f{fid}_pose"])
    noise = gio.sampler_noise(gio.S(g[f"f{fid}_seed"]), pts.shape[0])
    coord, label, weight, keep, n_all = DataSampler(cfg)._run(pts.cuda(), lpm, pose, noise)
    # oracle, dense: recompute the pieces of sample_region_specific without the final compaction
    lc = R.LocalCloud.empty(resolution=0.2, buffer_size=cfg.local_buffer_size, map_size=cfg.local_map_size)
    lc.buffer_pt_index = lpm.buffer_pt_index.cpu()
    lc.points = lpm.local_point_cloud_map.cpu()
    sc = R.SamplerConfig()
    xyz, disp, ratio, depth, disp_s, n_all_o = R._ray_samples(sc, pts, noise)
    assert n_all == n_all_o == 8
    Rn = pts.shape[0]
    n_surface = Rn * (sc.surface_sample_n + 1)
    d, ok = R.region_sdf(lc, R.transform(xyz[Rn:n_surface], pose))
    lab = -1 * disp.squeeze(1)
    lab[Rn:n_surface] = torch.where(disp_s.squeeze(1) < 0, 1, -1) * d
    kp = torch.ones(Rn * n_all, dtype=torch.bool)
    kp[Rn:n_surface] = ok
    w, _ = R._weights(sc, depth, ratio, Rn, n_all)
    w[n_surface:] *= -1.0
    want_c, want_l = R._ray_major(xyz, n_all).numpy(), R._ray_major(lab, n_all).numpy()
    want_w, want_k = R._ray_major(w.squeeze(1), n_all).numpy(), R._ray_major(kp, n_all).numpy()
    assert np.abs(coord.cpu().numpy() - want_c).max() <= 1e-4
    assert np.abs(weight.cpu().numpy() - want_w).max() <= 1e-6
    k = keep.bool().cpu().numpy()
    amb = R.region_sdf_ambiguity(lc, R.transform(xyz[Rn:n_surface], pose))
    nm_ = torch.zeros(Rn * n_all, dtype=torch.bool)
    nm_[Rn:n_surface] = amb["any"]
    named = R._ray_major(nm_, n_all).numpy()
    assert not ((k != want_k) & ~named).any() and (k != want_k).mean() <= 1e-3   # masks differ only where the oracle names the sample
    both = k & want_k
    err = np.abs(label.cpu().numpy() - want_l)
    assert err[both & ~named].max() <= 1e-4, (int((err[both & ~named] > 1e-4).sum()), float(err[both & ~named].max()))
    assert named.mean() <= 1e-2 and np.median(err[both]) <= 2e-6
    assert (~k).sum() > 0 and k.sum() > 0


def test_projective_sampler_kernel_g9():
    from clid_slam_amd import DataSampler

    g = gio.load("g9_sampler.npz")
    cfg = _cfg(g)
    pts = gio.T(g["f0_points"])
    noise = gio.sampler_noise(gio.S(g["pin_seed"]), pts.shape[0])
    coord, label, normal, sem, color, weight = DataSampler(cfg).sample_pin(pts.cuda(), None, None, None, noise=noise)
    assert normal is None and sem is None and color is None
    assert np.abs(coord.cpu().numpy() - g["pin_coord"]).max() <= 1e-4
    # free-space labels reach tens of metres: (ratio - 1) * |ray| carries the 1-ulp difference of |ray|
    assert (np.abs(label.cpu().numpy() - g["pin_label"]) <= 2e-6 + 1e-6 * np.abs(g["pin_label"])).all()
    assert np.abs(weight.cpu().numpy() - g["pin_weight"]).max() <= 1e-6
    # behind-surface drop-off variant against the oracle
    cfg.behind_dropoff_on = True
    c2, l2, _, _, _, w2 = DataSampler(cfg).sample_pin(pts.cuda(), None, None, None, noise=noise)
    oc, ol, ow = R.sample_projective(R.SamplerConfig(behind_dropoff_on=True), pts, noise)
    assert np.abs(w2.cpu().numpy() - ow.numpy()).max() <= 5e-6  # inherits the displacement's rounding through (hi - disp) / 0.96
    assert (np.abs(l2.cpu().numpy() - ol.numpy()) <= 2e-6 + 1e-6 * np.abs(ol.numpy())).all()


def test_device_rng_contract():
    """Without injected noise the sampler consumes torch's device generator in the reference's order."""
    from clid_slam_amd import DataSampler

    g = gio.load("g9_sampler.npz")
    cfg = _cfg(g)
    pts = gio.T(g["f0_points"]).cuda()
    Rn = pts.shape[0]
    torch.manual_seed(5)
    a = DataSampler(cfg).sample_pin(pts)
    torch.manual_seed(5)
    noise = (torch.randn(Rn * 4, 1, device="cuda"), torch.rand(Rn * 2, 1, device="cuda"), torch.rand(Rn * 1, 1, device="cuda"))
    b = DataSampler(cfg).sample_pin(pts, noise=noise)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[5], b[5])


def test_dropped_predraw_does_not_shift_the_random_stream():
    """A predraw for another ray count is dropped AND the generator goes back to where it was: the draws that follow are
    the ones a run without the predraw makes (the reference's stream; identical on every rank of a data-parallel group)."""
    from clid_slam_amd import DataSampler

    g = gio.load("g9_sampler.npz")
    cfg = _cfg(g)
    pts = gio.T(g["f0_points"]).cuda()
    torch.manual_seed(5)
    a = DataSampler(cfg).sample_pin(pts)
    after_a = torch.rand(4, device="cuda")
    torch.manual_seed(5)
    smp = DataSampler(cfg)
    smp.predraw(pts.shape[0] + 17, pts.device)
    b = smp.sample_pin(pts)
    after_b = torch.rand(4, device="cuda")
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[5], b[5]) and torch.equal(after_a, after_b)
    torch.manual_seed(5)
    smp.predraw(pts.shape[0], pts.device)  # the matching predraw is consumed as it is
    c = smp.sample_pin(pts)
    assert torch.equal(a[0], c[0]) and torch.equal(after_a, torch.rand(4, device="cuda"))


def test_process_frame_g10():
    """Mapper.process_frame over the three G10 frames: pool sizes, per-frame membership, label / coordinate
    sums, the new-sample selection and the neural-point map growth against the reference's own run."""
    from clid_slam_amd import Decoder, LocalPointCloudMap, Mapper, NeuralPoints

    g9, g = gio.load("g9_sampler.npz"), gio.load("g10_process_frame.npz")
    cfg = _cfg(g, buffer_size=int(gio.S(g["buffer_size"])))
    cfg.window_radius = float(gio.S(g["window_radius"]))
    cfg.local_map_radius = float(gio.S(g["local_map_radius"]))

    class DS:
        lose_track = False
        stop_status = False
        processed_frame = 0
        gt_pose_provided = True
        gt_poses = np.stack([g9[f"f{i}_pose"] for i in range(3)])

    torch.manual_seed(42)
    nm = NeuralPoints(cfg)
    nm.travel_dist = gio.T(g["travel_dist"]).cuda()
    dec = Decoder(cfg, cfg.geo_mlp_hidden_dim, cfg.geo_mlp_level, 1)
    ds = DS()
    mp = Mapper(cfg, ds, nm, LocalPointCloudMap(cfg), dec)

    class Replay:  # the reference's draws for this frame, injected into the sampler
        def __init__(self, inner):
            self.inner, self.noise = inner, None

        def sample(self, pts, lpm, pose):
            return self.inner.sample(pts, lpm, pose, noise=self.noise)

        def _run(self, pts, lpm, pose, noise):  # the fused compaction path of process_frame enters here
            return self.inner._run(pts, lpm, pose, self.noise)

    from clid_slam_amd import DataSampler

    mp.sampler = Replay(DataSampler(cfg))
    for fid in range(3):
        pts, pose = gio.T(g9[f"f{fid}_points"]).cuda(), gio.T(g9[f"f{fid}_pose"]).cuda()
        ds.processed_frame = fid
        if fid > 0:
            nm.point_certainties[nm.neural_points[:, 0] < 1.0] = 2.0
        mp.sampler.noise = gio.sampler_noise(gio.S(g[f"f{fid}_seed"]), pts.shape[0])
        mp.process_frame(pts, None, pose, fid)
        want_pool, want_cur = int(gio.S(g[f"f{fid}_pool_count"])), int(gio.S(g[f"f{fid}_cur_count"]))
        assert abs(mp.pool_sample_count - want_pool) <= 3 and abs(mp.cur_sample_count - want_cur) <= 3
        hist = torch.bincount(mp.time_pool.long(), minlength=3).cpu().numpy()
        assert np.abs(hist - g[f"f{fid}_time_hist"]).max() <= 3
        assert abs(float(mp.sdf_label_pool.double().sum()) - float(gio.S(g[f"f{fid}_label_sum"]))) <= 0.05
        assert np.abs(mp.global_coord_pool.double().sum(0).cpu().numpy() - g[f"f{fid}_gcoord_sum"]).max() <= 0.02 * 60
        # the map grows from samples with |label| < 0.125: a label within rounding of that bound moves a point
        assert abs(int(nm.count()) - int(gio.S(g[f"f{fid}_n_points"]))) <= 5
        assert abs(int(nm.local_count()) - int(gio.S(g[f"f{fid}_n_local"]))) <= 5
        assert abs(mp.new_idx.shape[0] - g[f"f{fid}_new_idx"].shape[0]) <= 5
        assert mp.adaptive_iter_offset == int(gio.S(g[f"f{fid}_iter_offset"]))
        assert mp.global_coord_pool.shape[0] == mp.sdf_label_pool.shape[0] == mp.weight_pool.shape[0] == mp.time_pool.shape[0]
    if mp.pool_sample_count == g["final_label"].shape[0]:
        assert np.abs(mp.global_coord_pool.cpu().numpy() - g["final_global_coord"]).max() <= 1e-4
        assert np.array_equal(mp.time_pool.cpu().numpy(), g["final_time"])
        err = np.abs(mp.sdf_label_pool.cpu().numpy() - g["final_label"])
        assert (err > 1e-4).mean() <= 1e-3
    # and the loop trains on the pool it just built
    mp.mapping(3)
    assert torch.isfinite(mp.last_losses).all()


def test_process_frame_fused_glue_equals_the_op_chain(monkeypatch):
    """process_frame's fused glue (clid_sample_compact: compaction + stamps + world coordinates + near-surface subset;
    clid_new_sample_select: certainty probe + tests + index list) against the torch op chain it replaces
    (utils/mapper.py:240-283, :297-310, :400-423), frame by frame on the sequence workload: identical pools, maps and
    selections."""
    import bench_sequence as BS

    runs = []
    for fused in ("0", "1"):
        monkeypatch.setenv("CLID_FUSED_COMPACT", fused)
        monkeypatch.setenv("CLID_FUSED_NEWSEL", fused)
        torch.manual_seed(0)
        snaps = []

        def grab(mp, nm):
            snaps.append((mp.coord_pool.clone(), mp.global_coord_pool.clone(), mp.sdf_label_pool.clone(), mp.weight_pool.clone(),
                          mp.time_pool.clone(), mp.new_idx.clone(), mp.adaptive_iter_offset, mp.cur_sample_count,
                          nm.neural_points.clone(), nm.local_neural_points.clone()))

        cfg, rows, checks, objs = BS.run(3, "cuda:0", quiet=True, after_process=grab)
        runs.append(snaps)
    assert len(runs[0]) == len(runs[1]) == 3
    for a, b in zip(*runs):
        for x, y in zip(a, b):
            if isinstance(x, torch.Tensor):
                assert x.shape == y.shape and torch.equal(x, y)
            else:
                assert x == y
    assert runs[1][-1][5].numel() > 0 and runs[1][-1][0].shape[0] > 100_000


def test_process_frame_without_voxel_round_trips_equals_the_two_phase_path(monkeypatch):
    """The two voxel passes of a frame left in flight (CLID_ASYNC_VOXEL, default: their index lists and counts are consumed
    on the device by the raw-point map update and by the insert + window; counts come back with the reads that follow)
    against the paths that read each voxel count back: identical raw-point maps, pools, neural-point maps, windows and
    selections on six frames of the sequence workload."""
    import bench_sequence as BS
    from clid_slam_amd import _lib

    runs, reads = [], []
    for mode in ("0", "1"):
        monkeypatch.setenv("CLID_ASYNC_VOXEL", mode)
        torch.manual_seed(0)
        snaps, n_reads = [], [0]
        orig = _lib.load().clid_read_back

        def grab(mp, nm):
            lpm = mp.local_point_cloud_map
            snaps.append((mp.coord_pool.clone(), mp.global_coord_pool.clone(), mp.sdf_label_pool.clone(), mp.weight_pool.clone(),
                          mp.time_pool.clone(), mp.new_idx.clone(), mp.adaptive_iter_offset, mp.cur_sample_count, mp.cur_new_point_ratio,
                          nm.neural_points.clone(), nm.local_neural_points.clone(), nm.buffer_pt_index.clone(), nm.global2local.clone(),
                          nm.local_mask.clone(), lpm.local_point_cloud_map.clone(), lpm.buffer_pt_index.clone()))

        cfg, rows, checks, objs = BS.run(6, "cuda:0", quiet=True, after_process=grab)
        runs.append(snaps)
    assert len(runs[0]) == len(runs[1]) == 6
    for a, b in zip(*runs):
        for x, y in zip(a, b):
            if isinstance(x, torch.Tensor):
                assert x.shape == y.shape and torch.equal(x, y)
            else:
                assert x == y


def test_process_frame_with_the_gated_pool_compaction_equals_the_default(monkeypatch):
    """CLID_POOL_GATE=1 (clid_pool_filter_after: the five-array compaction held back behind the map growth's voxel pass by an
    event; the passes in front of it are not) is a pure change of schedule: identical pools, maps, windows and selections."""
    import bench_sequence as BS

    runs = []
    for mode in ("0", "1"):
        monkeypatch.setenv("CLID_POOL_GATE", mode)
        torch.manual_seed(0)
        snaps = []

        def grab(mp, nm):
            snaps.append((mp.coord_pool.clone(), mp.global_coord_pool.clone(), mp.sdf_label_pool.clone(), mp.weight_pool.clone(),
                          mp.time_pool.clone(), mp.new_idx.clone(), mp.adaptive_iter_offset, mp.cur_sample_count,
                          nm.neural_points.clone(), nm.local_neural_points.clone(), nm.global2local.clone()))

        BS.run(5, "cuda:0", quiet=True, after_process=grab)
        runs.append(snaps)
    assert len(runs[0]) == len(runs[1]) == 5
    for a, b in zip(*runs):
        for x, y in zip(a, b):
            if isinstance(x, torch.Tensor):
                assert x.shape == y.shape and torch.equal(x, y)
            else:
                assert x == y
