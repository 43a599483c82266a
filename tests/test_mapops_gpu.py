"""GPU: map-maintenance kernels ("next" row N4) against the torch / oracle restatements of the same rules."""
import numpy as np
import pytest
import torch

import golden_io as gio
from oracle import sampler_ref as R

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,voxel,seed", [(1, 0.2, 0), (7, 0.4, 1), (5000, 0.2, 2), (200000, 0.4, 3), (450000, 0.1, 4),
                                          (1_500_000, 0.08, 5)])  # the last one overflows the ordering buckets: library sort
def test_voxel_down_sample_kernel_matches_the_reference_rule(n, voxel, seed):
    """clid_voxel_down_sample == the oracle's restatement of utils/tools.py:639-682 (scatter-amin on the CPU) on
    identical points: same indices in the same order, including the stride-aliasing quirk."""
    from clid_slam_amd.tools import voxel_down_sample_torch

    g = torch.Generator().manual_seed(seed)
    pts = (torch.rand((n, 3), generator=g) - 0.3) * torch.tensor([40.0, 25.0, 6.0])
    if n > 10:
        pts[n // 2] = pts[n // 3]  # exact duplicates: the lower index wins
    want = R.voxel_down_sample(pts, voxel)
    got = voxel_down_sample_torch(pts.cuda(), voxel)
    assert got.dtype == torch.int64 and got.is_cuda
    assert torch.equal(got.cpu(), want)


def _voxel_async(pts, voxel, n_live=None):
    from clid_slam_amd.tools import voxel_down_sample_async

    counts = torch.full((2,), -7, dtype=torch.int64, device="cuda")
    n_dev = None if n_live is None else torch.tensor([n_live], dtype=torch.int64, device="cuda")
    idx = voxel_down_sample_async(pts, voxel, counts, n_dev=n_dev)
    m, bad = counts.tolist()
    return idx[:m].cpu(), m, bad


@pytest.mark.parametrize("n,voxel,seed", [(1, 0.2, 0), (7, 0.4, 1), (5000, 0.2, 2), (200000, 0.4, 3), (450000, 0.1, 4), (1_500_000, 0.08, 5)])
def test_voxel_down_sample_without_a_round_trip(n, voxel, seed):
    """clid_voxel_down_sample_async (index list + count left on the device; buckets beyond their capacity ranked by counting
    on the device instead of by the library sort, which needs the count on the host -- the last case has ~1e6 voxels in 512
    buckets of 4096) == the oracle's rule; and on a prefix given by a device-side count."""
    g = torch.Generator().manual_seed(seed)
    pts = (torch.rand((n, 3), generator=g) - 0.3) * torch.tensor([40.0, 25.0, 6.0])
    if n > 10:
        pts[n // 2] = pts[n // 3]
    got, m, bad = _voxel_async(pts.cuda(), voxel)
    assert bad == 0 and torch.equal(got, R.voxel_down_sample(pts, voxel))
    if 10 < n <= 450000:
        live = (2 * n) // 3
        got, m, bad = _voxel_async(pts.cuda(), voxel, n_live=live)
        assert bad == 0 and torch.equal(got, R.voxel_down_sample(pts[:live], voxel))


def test_voxel_down_sample_without_a_round_trip_skewed_buckets():
    """Every sampled point (the splitters come from 1024 evenly spaced inputs) in ONE voxel, 30000 other voxels beyond it: the
    last bucket receives all of them (capacity 4096), the rest goes through the spill list and the counting ranks."""
    n = 40960
    g = torch.Generator().manual_seed(11)
    pts = torch.rand((n, 3), generator=g) * torch.tensor([60.0, 50.0, 10.0]) + torch.tensor([5.0, 5.0, 5.0])
    pts[torch.arange(0, n, n // 1024)] = torch.tensor([0.05, 0.05, 0.05])
    got, m, bad = _voxel_async(pts.cuda(), 0.5)
    want = R.voxel_down_sample(pts, 0.5)
    assert bad == 0 and m == want.numel() and m > 20000 and torch.equal(got, want)


def test_voxel_down_sample_kernel_on_a_scan():
    from clid_slam_amd.synth import box_room_scan
    from clid_slam_amd.tools import voxel_down_sample_torch

    scan = box_room_scan(n_elev=64, n_azim=1024, seed=5, vox_down_m=0.0) + torch.tensor([3.0, -2.0, 1.5])
    for voxel in (0.1, 0.2, 0.4):
        want = R.voxel_down_sample(scan, voxel)
        got = voxel_down_sample_torch(scan.cuda(), voxel).cpu()
        assert torch.equal(got, want)
        # one point per occupied voxel of the reference's own linearisation
        assert got.unique().numel() == got.numel()


def test_map_growth_on_the_gpu_rebuilds_the_reference_map():
    """NeuralPoints.update / reset_local_map on the device (voxel kernel + deterministic last-writer slot updates)
    against G7, the map the reference builds single-threaded from the same three frames: identical points, stamps,
    slot table and local window."""
    from clid_slam_amd import HotPathConfig, NeuralPoints
    from clid_slam_amd.synth import box_room_pool

    z = gio.load("g7_mapbuild.npz")
    cfg = HotPathConfig()
    cfg.device = "cuda"
    cfg.buffer_size = int(gio.S(z["buffer_size"]))
    torch.manual_seed(42)
    nm = NeuralPoints(cfg)
    nm.travel_dist = torch.tensor([0.0, 400.0, 403.5], device="cuda")
    sensors = [(0.0, 0.0, 1.5), (6.0, 2.0, 1.5), (9.0, 3.0, 1.6)]
    for fid, s in enumerate(sensors):
        d = box_room_pool(cfg, n_elev=32, n_azim=256, seed=42 + fid, sensor=s)
        near = d["sdf_label"].abs() < cfg.surface_sample_range_m * 0.5
        nm.update(d["coord"][near].cuda(), d["sensor"].cuda(), torch.eye(3, device="cuda"), fid)
    assert nm.count() == z["neural_points"].shape[0]
    assert np.array_equal(nm.neural_points.cpu().numpy(), z["neural_points"])
    assert np.array_equal(nm.point_ts_create.cpu().numpy(), z["point_ts_create"])
    occ = torch.nonzero(nm.buffer_pt_index >= 0).flatten()
    assert np.array_equal(occ.cpu().numpy(), z["table_slot"])
    assert np.array_equal(nm.buffer_pt_index[occ].cpu().numpy(), z["table_idx"])
    nm.local_map_radius = 12.0
    nm.reset_local_map(torch.tensor(sensors[-1], device="cuda"), torch.eye(3, device="cuda"), 2, reboot_map=True)
    assert np.array_equal(nm.global2local.cpu().numpy(), z["global2local"])
    assert np.array_equal(nm.local_neural_points.cpu().numpy(), z["local_neural_points"])


def test_prune_and_recreate_hash_on_the_gpu_match_the_reference():
    from test_host_logic import _g11_run

    _g11_run("cuda")


def _random_map(n, buffer_size, seed):
    from clid_slam_amd import NeuralPoints
    from clid_slam_amd.config import HotPathConfig

    cfg = HotPathConfig()
    cfg.device = "cuda"
    cfg.buffer_size = buffer_size
    nm = NeuralPoints(cfg)
    g = torch.Generator().manual_seed(seed)
    nm.neural_points = ((torch.rand((n, 3), generator=g) - 0.5) * torch.tensor([60.0, 40.0, 6.0])).cuda()
    nm.point_orientations = torch.nn.functional.normalize(torch.randn((n, 4), generator=g), dim=1).cuda()
    nm.point_ts_create = torch.randint(0, 40, (n,), generator=g, dtype=torch.int32).cuda()
    nm.point_ts_update = (nm.point_ts_create + torch.randint(0, 10, (n,), generator=g, dtype=torch.int32).cuda()).int()
    nm.point_certainties = (torch.rand(n, generator=g) * 4.0).cuda()
    nm.geo_features = torch.cat((torch.randn((n, 8), generator=g), torch.zeros(1, 8)), 0).cuda()
    nm.travel_dist = (torch.arange(64, dtype=torch.float32) * 1.7).cuda()
    nm.cur_ts = 49
    nm.diff_travel_dist_local = 30.0
    return nm


_MAP_ARRAYS = ("neural_points", "point_orientations", "point_ts_create", "point_ts_update", "point_certainties", "geo_features",
               "buffer_pt_index")


@pytest.mark.parametrize("n,buffer_size", [(40_000, 1 << 22), (25_000, 4093), (3, 1 << 16)])
@pytest.mark.parametrize("global_prune", [False, True])
def test_prune_and_recreate_hash_kernels_vs_the_torch_chain(monkeypatch, n, buffer_size, global_prune):
    """NeuralPoints.prune_map / recreate_hash (model/neural_points.py:771-812, 840-929) through clid_map_prune_select /
    clid_map_gather / clid_voxel_down_sample_min_value / clid_map_rehash against the same methods as torch op chains
    (CLID_FUSED_MAINTENANCE=0; pinned on the reference by G11): every array and the whole slot table bit for bit, incl. a
    table of 4093 slots where most slots are named by several points (last writer stays)."""
    got = {}
    for fused in ("1", "0"):
        monkeypatch.setenv("CLID_FUSED_MAINTENANCE", fused)
        nm = _random_map(n, buffer_size, seed=n)
        out = [nm.prune_map(1.5, min_prune_count=2 if n > 3 else 0, global_prune=global_prune)]
        nm.recreate_hash(None, None, True, True, 49)
        out += [getattr(nm, k).clone() for k in _MAP_ARRAYS]
        nm.recreate_hash(None, None, True, False, 49)
        out += [nm.buffer_pt_index.clone()]
        nm.recreate_hash(torch.zeros(3, device="cuda"), torch.eye(3, device="cuda"), False, False, 49)
        out += [getattr(nm, k).clone() for k in _MAP_ARRAYS] + [nm.local_neural_points.clone(), nm.global2local.clone()]
        got[fused] = out
    assert got["1"][0] == got["0"][0] and (n <= 3 or got["1"][0] is True)
    assert len(got["1"]) == len(got["0"])
    for a, b in zip(got["1"][1:], got["0"][1:]):
        assert a.shape == b.shape and a.dtype == b.dtype and torch.equal(a, b)
    m = got["1"][1].shape[0]
    assert got["1"][6].shape[0] == m + 1 and (n <= 3 or m < n)


def test_local_window_repeats_when_the_local_map_outgrew_its_capacity():
    """reset_local_map sizes the local arrays from the previous window (not from the whole map); a window that grew past that
    capacity is detected from the device-side count and gathered again: same result as with room for everything."""
    got = []
    for last in (None, 10, 39_000):
        nm = _random_map(40_000, 1 << 22, seed=7)
        nm.recreate_hash(None, None, True, True, 49)
        nm.local_map_radius = 1000.0
        nm.diff_travel_dist_local = 1e9
        nm._last_local_m = last
        nm.reset_local_map(torch.zeros(3, device="cuda"), torch.eye(3, device="cuda"), 49)
        assert nm.local_neural_points.shape[0] == 40_000 and nm._last_local_m == 40_000
        got.append([t.clone() for t in (nm.local_neural_points, nm.local_point_orientations, nm.local_point_certainties,
                                        nm.local_point_ts_update, nm.local_geo_features.data, nm.global2local, nm.local_mask)])
        if last is not None:  # capacity = min(n, 1.25 * last + 16384): 16 396 rows the first time, everything the second
            assert nm.local_neural_points.untyped_storage().nbytes() <= 40_000 * 12
    for other in got[1:]:
        for a, b in zip(got[0], other):
            assert torch.equal(a, b)


def test_prune_map_below_the_minimum_count_changes_nothing():
    nm = _random_map(5000, 1 << 20, seed=5)
    before = [getattr(nm, k).clone() for k in _MAP_ARRAYS[:-1]]
    assert nm.prune_map(1.5, min_prune_count=5000) is False
    for k, b in zip(_MAP_ARRAYS[:-1], before):
        assert torch.equal(getattr(nm, k), b)


def _splitmix64(x):
    M = (1 << 64) - 1
    x = (x + 0x9E3779B97F4A7C15) & M
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & M
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & M
    return x ^ (x >> 31)


@pytest.mark.parametrize("n_a,n_b,capacity", [(50_000, 10_000, 30_000), (1000, 300, 10_000_000), (0, 5000, 2000), (70_001, 0, 1000)])
def test_pool_filter_window_capacity_and_order(n_a, n_b, capacity):
    """clid_pool_filter (utils/mapper.py:297-392): append, float64 window test, `kept - capacity` uniform picks WITH
    replacement among the kept samples dropped (:352-361), stable compaction of the five arrays; the two counts.  The drop
    uses a counter-based generator (splitmix64(seed + t) mod kept), restated here."""
    import ctypes as C

    from clid_slam_amd import _lib

    lib = _lib.load()
    dev = "cuda:0"
    g = torch.Generator().manual_seed(n_a + n_b)
    n = n_a + n_b
    gcoord = (torch.rand((n, 3), generator=g) * 40.0 - 20.0)
    coord = torch.rand((n, 3), generator=g)
    label, weight = torch.randn(n, generator=g), torch.rand(n, generator=g)
    stamp = torch.randint(0, 50, (n,), generator=g, dtype=torch.int32)
    origin = (1.5, -2.0, 0.5)
    radius = 15.0
    d = [t.to(dev).contiguous() for t in (coord, gcoord, label, weight, stamp)]
    a = [t[:n_a] for t in d]
    b = [t[n_a:] for t in d]
    out = [torch.empty((n + 8, 3), device=dev), torch.empty((n + 8, 3), device=dev), torch.empty(n + 8, device=dev),
           torch.empty(n + 8, device=dev), torch.empty(n + 8, device=dev, dtype=torch.int32)]
    ws = torch.empty(int(lib.clid_pool_workspace_bytes(n)), device=dev, dtype=torch.uint8)
    counts = torch.zeros(3, device=dev, dtype=torch.int64)
    seed = 0x1234ABCD
    ptr = lambda t: t.data_ptr() if t.numel() else None
    _lib.check(lib.clid_pool_filter(ptr(a[0]), ptr(a[1]), ptr(a[2]), ptr(a[3]), ptr(a[4]), n_a, ptr(b[0]), ptr(b[1]), ptr(b[2]),
                                    ptr(b[3]), ptr(b[4]), n_b, (C.c_double * 3)(*origin), radius * radius, capacity, seed,
                                    out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), out[3].data_ptr(), out[4].data_ptr(),
                                    counts.data_ptr(), ws.data_ptr(), None, _lib.stream()), "clid_pool_filter")
    kept, kept_cur = [int(v) for v in counts[:2].tolist()]
    # host restatement
    dist2 = ((gcoord.double() - torch.tensor(origin, dtype=torch.float64)) ** 2).sum(1)
    flag = (dist2 < radius * radius).numpy()
    win = np.nonzero(flag)[0]
    if n > capacity and len(win) > capacity:
        for t in range(len(win) - capacity):
            flag[win[_splitmix64(seed + t) % len(win)]] = False
    want = np.nonzero(flag)[0]
    assert kept == len(want) and kept_cur == int((want >= n_a).sum())
    if n > capacity and len(win) > capacity:
        assert capacity <= kept < len(win)
    for got, src in zip(out, (coord, gcoord, label, weight, stamp)):
        assert torch.equal(got[:kept].cpu(), src[want])


def test_an_outlier_that_defeats_the_device_side_voxel_ordering_is_recovered_from():
    """a scan with one point far outside (a bounding box beyond 2^17 voxels per axis) makes the device-side voxel
    ordering give up.  The pass then publishes ZERO voxels, so the consumers that size themselves by that count on the device
    (the raw-point map update, the neural-point insert) change nothing, and the host repeats the step through the pass with the
    library sort at its next read-back: the maps end up exactly as on the path that never left the voxel pass in flight."""
    from clid_slam_amd import HotPathConfig, LocalPointCloudMap, NeuralPoints
    from clid_slam_amd.synth import box_room_scan
    from clid_slam_amd.tools import voxel_down_sample_async

    dev = "cuda:0"
    cfg = HotPathConfig()
    cfg.device = dev
    pts = box_room_scan(n_elev=32, n_azim=512, seed=3).to(dev)
    pts = torch.cat((pts[:, :3].float(), torch.tensor([[60000.0, -3.0, 1.0]], device=dev))).contiguous()
    sensor = torch.zeros(3, device=dev, dtype=torch.float64)

    # ---- raw-point map: deferred counts + voxel pass in flight (what Mapper.process_frame arms) vs the plain path
    maps = []
    for armed in (False, True):
        lpm = LocalPointCloudMap(cfg)
        lpm.update_map(sensor, pts[:2000])        # a first frame through the plain path
        if armed:
            lpm._defer_counts = torch.zeros(2, device=dev, dtype=torch.int64)
            lpm._defer_vox = torch.zeros(2, device=dev, dtype=torch.int64)
        lpm.update_map(sensor, pts)
        if armed:
            assert lpm._count_pending
            lpm._finish_count()
            assert getattr(lpm, "vox_fallbacks", 0) == 1
        maps.append((lpm.local_point_cloud_map.clone(), lpm.buffer_pt_index.clone()))
    assert maps[0][0].shape == maps[1][0].shape and torch.equal(maps[0][0], maps[1][0]) and torch.equal(maps[0][1], maps[1][1])

    # ---- neural-point map: insert + window on the voxel pass' device-side list vs the plain path
    res = []
    for armed in (False, True):
        torch.manual_seed(0)
        nm = NeuralPoints(cfg)
        nm.travel_dist = torch.zeros(4, device=dev)
        nm.update(pts[:3000], torch.zeros(3, device=dev), torch.eye(3, device=dev), 0)
        assert nm.update_is_fused()
        if armed:
            counts = nm.update_counts(dev)
            idx = voxel_down_sample_async(pts, nm.resolution, counts[3:5])
            nm._presampled = (pts, (idx, counts[3:5], None))
        ratio = nm.update(pts, torch.zeros(3, device=dev), torch.eye(3, device=dev), 1)
        if armed:
            assert getattr(nm, "vox_fallbacks", 0) == 1
        res.append((ratio, nm.neural_points.clone(), nm.buffer_pt_index.clone(), nm.local_neural_points.clone(), nm.global2local.clone()))
    assert res[0][0] == res[1][0]
    for a, b in zip(res[0][1:], res[1][1:]):
        assert a.shape == b.shape and torch.equal(a, b)


@pytest.mark.parametrize("elem", [4, 8])
def test_exclusive_scan_vs_cumsum(elem):
    """The hand-written exclusive prefix sum behind every compaction of the map / pool maintenance (csrc/mapops.hip
    scan_exclusive: one launch up to 4 096 elements, two beyond) against numpy on the sizes around its tile and launch
    boundaries; flags (0 / 1) and wide values."""
    from clid_slam_amd import _lib

    lib = _lib.load()
    rng = np.random.default_rng(3)
    for n in (1, 5, 63, 64, 1023, 1024, 1025, 4095, 4096, 4097, 8192, 40959, 40960, 40961, 65536, 131072, 1_000_003, 2_500_001):
        for wide in (False, True):
            if elem == 4:
                host = rng.integers(0, 800 if wide else 2, n, dtype=np.int64).astype(np.int32)
                dev = torch.from_numpy(host).cuda()
            else:
                host = rng.integers(0, (1 << 40) if wide else 2, n, dtype=np.int64)
                dev = torch.from_numpy(host).cuda()
            out = torch.full_like(dev, -1)
            scratch = torch.empty(int(lib.clid_debug_scan_scratch_bytes(n)), device="cuda", dtype=torch.uint8)
            _lib.check(lib.clid_debug_scan(dev.data_ptr(), out.data_ptr(), n, elem, scratch.data_ptr(), _lib.stream()), "clid_debug_scan")
            want = np.concatenate(([0], np.cumsum(host.astype(np.int64))[:-1]))
            got = out.cpu().numpy().astype(np.int64)
            assert np.array_equal(got, want), (n, wide, int((got != want).sum()))
