"""Measurement of the two built "next" rows of SURVEY.md section 8(f) on one MI355X (bench.py keeps the
headline metric; this prints one JSON line per row):

  N1  tracking measurement model   IEKFOM.h_model / normal equations  (utils/error_state_iekf.py:176-305)
  N3  dense SDF inference          Mesher.query_points                (utils/mesher.py:38-163)
  N2  sample + label generation    DataSampler.sample / process_frame (utils/data_sampler.py:260-402,
                                   model/local_point_cloud_map.py:98-201, utils/mapper.py:159-470)

Both run on the same synthetic box-room map as bench.py after a short training run, with inputs resident in
HBM, timed with events on the launch stream.  `roofline.achieved` = algorithmic bytes per launch / average
launch time (bytes per query point: 688 B search + 216 B feature gather + outputs).
The CPU leg times the oracle on a bounded sample (test infrastructure used only as the reported baseline).

    python bench_next.py [--points 4194304] [--track-points 8192] [--no-cpu-baseline]
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
HBM_PEAK_GBS = 8000.0
BYTES_SEARCH = 688.0
BYTES_FEAT = 216.0


def pmc_traffic(kernel, default_workload: bool):
    """HBM bytes per launch of `kernel` from the committed counter passes (tools/next_rows_traffic.sh; default workload only)."""
    if not default_workload:
        return None
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "r04_next_rows_traffic.json")))
        return next(v["traffic_bytes"] for k, v in tj.items() if isinstance(v, dict) and kernel in k)
    except (OSError, StopIteration, KeyError, ValueError):
        return None


def timed(fn, reps, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps


def timed_device(fn, reps, warm=3):
    """Seconds of DEVICE time per call: the launches queue up behind a spin kernel, so the span between the two events is the
    kernels back to back -- what a caller that iterates the model (5-20 evaluations per scan) sees once its host side keeps up,
    and what `timed` hides when the Python glue of one call (view, ctypes, a 28-double fill) is longer than the kernel."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(int(2.0e7))  # ~10 ms of spinning: time to enqueue everything below
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps


def sampler_row(cfg, args):
    """N2: one Ouster-128-like scan (box room) through the raw-point map update, the fused sampler launch and
    the whole Mapper.process_frame."""
    from clid_slam_amd import DataSampler, Decoder, LocalPointCloudMap, Mapper, NeuralPoints
    from clid_slam_amd.synth import box_room_scan
    from clid_slam_amd.tools import transform_torch

    dev = "cuda:0"
    sensor = (0.0, 0.0, 1.5)
    scan = box_room_scan(sensor=sensor).to(dev)  # sensor frame, ~1e5 rays after the 0.1 m voxel filter
    pose = torch.eye(4, dtype=torch.float64, device=dev)
    pose[:3, 3] = torch.tensor(sensor, dtype=torch.float64)
    lpm = LocalPointCloudMap(cfg)
    lpm.update_map(pose[:3, 3], transform_torch(scan, pose))
    smp = DataSampler(cfg)
    R = scan.shape[0]
    n_all = 1 + cfg.surface_sample_n + cfg.free_front_n + cfg.free_behind_n
    noise = (torch.randn(R * cfg.surface_sample_n, 1, device=dev), torch.rand(R * cfg.free_front_n, 1, device=dev),
             torch.rand(R * cfg.free_behind_n, 1, device=dev))
    t_kernel = timed(lambda: smp._run(scan, lpm, pose, noise), 50, warm=5)
    t_sample = timed(lambda: smp.sample(scan, lpm, pose), 20, warm=3)
    t_update = timed(lambda: lpm.update_map(pose[:3, 3], transform_torch(scan, pose)), 10, warm=2)

    class _DS:
        lose_track = False
        stop_status = False
        processed_frame = 0
        gt_pose_provided = False

    def one_frame():
        nm = NeuralPoints(cfg)
        nm.travel_dist = torch.zeros(4, device=dev)
        mp = Mapper(cfg, _DS(), nm, LocalPointCloudMap(cfg), Decoder(cfg, cfg.geo_mlp_hidden_dim, cfg.geo_mlp_level, 1))
        mp.process_frame(scan, None, pose, 0)
        return mp

    t_frame = timed(one_frame, 5, warm=1)
    mp = one_frame()
    # per ray: 12 B in + 28 B noise + 8 x 21 B out + 4 near-surface samples x 7 probes x (8 B slot + 12 B point)
    alg = R * (12.0 + 4.0 * (n_all - 1) + n_all * 21.0 + cfg.surface_sample_n * lpm.neighbor_idx.shape[0] * 20.0)
    line = {
        "row": "N2", "metric": "training samples/sec generated + labelled (DataSampler.sample, region-specific SDF)",
        "value": R * n_all / t_kernel, "unit": "samples/s", "n_gpus": 1, "dtype": "f32", "data": "synthetic",
        "us_fused_launch": 1e6 * t_kernel, "us_sample_with_compaction": 1e6 * t_sample,
        "us_raw_point_map_update": 1e6 * t_update, "ms_process_frame_first_frame": 1e3 * t_frame,
        "config": {"workload": "one box-room Ouster-128 scan, run_ncd128 sampler settings", "rays": R,
                   "samples_per_ray": n_all, "raw_map_points": int(lpm.local_point_cloud_map.shape[0]),
                   "pool_after_frame": int(mp.pool_sample_count), "neural_points": int(mp.neural_points.count())},
        "roofline": {"bound": "hbm", "kernel": "k_sample_frame", "achieved": alg / t_kernel / 1e9, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": alg / t_kernel / 1e9 / HBM_PEAK_GBS, "traffic": pmc_traffic("k_sample_frame", True),
                     "algorithmic_bytes_per_launch": alg},
    }
    if not args.no_cpu_baseline:
        from oracle import sampler_ref as SR

        threads = min(16, os.cpu_count() or 1)
        torch.set_num_threads(threads)
        lc = SR.LocalCloud.empty(resolution=cfg.local_voxel_size_m, buffer_size=cfg.local_buffer_size,
                                 map_size=cfg.local_map_size)
        lc.buffer_pt_index, lc.points = lpm.buffer_pt_index.cpu(), lpm.local_point_cloud_map.cpu()
        sc = SR.SamplerConfig(surface_sample_range_m=cfg.surface_sample_range_m, surface_sample_n=cfg.surface_sample_n,
                              free_behind_n=cfg.free_behind_n, free_front_n=cfg.free_front_n, max_range=cfg.max_range)
        pts_c, pose_c, noise_c = scan.cpu(), pose.cpu(), tuple(t.cpu() for t in noise)
        t0 = time.perf_counter()
        reps = 0
        while time.perf_counter() - t0 < 8.0:
            SR.sample_region_specific(sc, pts_c, lc, pose_c, noise_c)
            reps += 1
        dt = time.perf_counter() - t0
        line["cpu_baseline"] = {"value": R * n_all * reps / dt, "unit": "samples/s", "cores": threads, "kind": "port",
                                "sample": f"{reps} scans through the CPU oracle's sampler, {dt:.1f} s"}
    return line


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=4194304, help="dense query points (N3)")
    ap.add_argument("--track-points", type=int, default=8192, help="scan points per h_model call (N1)")
    ap.add_argument("--train-iters", type=int, default=100)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import bench
    from clid_slam_amd import HotPathConfig, _lib
    from clid_slam_amd import mesher, tracking

    _lib.load()
    cfg = HotPathConfig()
    cfg.device = "cuda:0"
    torch.cuda.set_device(0)
    nm, dec, mp, scene = bench.build_scene(cfg, "cuda:0")
    mp.mapping(args.train_iters)
    torch.cuda.synchronize()
    dev = "cuda:0"
    gen = torch.Generator().manual_seed(7)

    # ---- N1: one evaluation of the measurement model on a scan around the sensor
    near = scene["sdf_label"].abs() < 0.02
    surf = scene["coord"][near]
    sel = torch.randperm(surf.shape[0], generator=gen)[: args.track_points]
    sensor = scene["sensor"].to(torch.float32)
    pc_imu = (surf[sel] - sensor).to(dev).contiguous()
    rot = torch.eye(3)
    n1 = pc_imu.shape[0]
    t1a = timed(lambda: tracking._launch(nm, dec, cfg, rot, sensor, pc_imu, False, True), 200, warm=50)
    t1b = timed(lambda: tracking._launch(nm, dec, cfg, rot, sensor, pc_imu, True, False), 200, warm=10)
    rot_d, pos_d = rot.to(dev), sensor.to(dev)  # the filter's state as device tensors (utils/error_state_iekf.py:176-186)
    t1c = timed_device(lambda: tracking._launch(nm, dec, cfg, rot_d, pos_d, pc_imu, False, True), 100, warm=10)
    t1d = timed_device(lambda: tracking._launch(nm, dec, cfg, rot_d, pos_d, pc_imu, True, False), 100, warm=10)
    bound = tracking.bind(nm, dec, cfg, pc_imu)   # what the cached binding holds: one argument block per scan
    t1e = timed(lambda: bound.launch(rot, sensor, False, True), 200, warm=50)
    S, b, n_valid = tracking.normal_equations(nm, dec, cfg, rot, sensor, pc_imu)

    def full_ne():
        tracking.normal_equations(nm, dec, cfg, rot, sensor, pc_imu, host=True)
    for _ in range(20):
        full_ne()
    t0 = time.perf_counter()
    for _ in range(200):
        full_ne()
    t1f = (time.perf_counter() - t0) / 200
    def time_h_model(fused):  # the drop-in call the reference's update_iterated makes (host-synchronous: its outputs are sized by data)
        tracking._FUSED_ROWS = fused
        for _ in range(10):
            tracking.h_model(nm, dec, cfg, rot, sensor, pc_imu)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(100):
            tracking.h_model(nm, dec, cfg, rot, sensor, pc_imu)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / 100
    t1h_torch, t1h = time_h_model(False), time_h_model(True)
    alg1 = n1 * (12.0 + BYTES_SEARCH + BYTES_FEAT + 4.0)
    line1 = {
        "row": "N1", "metric": "tracking measurement-model points/sec (IEKFOM.h_model, fused normal equations)",
        "value": n1 / t1a, "unit": "points/s", "n_gpus": 1, "us_per_call_normal_equations": 1e6 * t1a,
        "us_h_model_call": 1e6 * t1h, "us_h_model_call_torch_glue": 1e6 * t1h_torch,
        "us_per_call_bound_object": 1e6 * t1e, "us_normal_equations_round_trip_host_result": 1e6 * t1f,
        "us_per_call_per_point_outputs": 1e6 * t1b, "us_per_call_device": 1e6 * t1c, "us_per_call_device_per_point_outputs": 1e6 * t1d, "dtype": "f32 (f64 reduction)", "data": "synthetic",
        "config": {"workload": "box-room map of bench.py, local map view, one IEKF iteration", "points": n1,
                   "valid_points": n_valid},
        "roofline": {"bound": "hbm", "kernel": "k_track_model", "achieved": alg1 / t1a / 1e9, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": alg1 / t1a / 1e9 / HBM_PEAK_GBS, "traffic": pmc_traffic("k_track_model", args.track_points == 8192),
                     "note": "value / frac: wall clock per Python call through the cached binding (key check + pose + one ctypes call: "
                             "host-bound; the launch clears the next call's reduction buffer itself); us_per_call_bound_object: the "
                             "same without the cache lookup; us_normal_equations_round_trip_host_result: launch + one-block finish "
                             "launch + poll of the pinned result block, per call, host-synchronous; us_per_call_device: the launches "
                             "back to back on the device, pose read from device tensors"},
    }

    # ---- N3: dense inference over points scattered through the mapped volume
    pool = scene["coord"]
    pick = torch.randint(0, pool.shape[0], (args.points,), generator=gen)
    x = (pool[pick] + 0.05 * torch.randn((args.points, 3), generator=gen)).to(dev).contiguous()
    run3 = lambda: mesher.query_points(nm, dec, cfg, x, query_locally=False)
    t3 = timed(run3, 5, warm=2)
    sdf3, _, _, mask3 = run3()
    alg3 = args.points * (12.0 + BYTES_SEARCH + BYTES_FEAT + 8.0)
    line3 = {
        "row": "N3", "metric": "dense SDF query points/sec (Mesher.query_points)", "value": args.points / t3,
        "unit": "points/s", "n_gpus": 1, "ms_per_call": 1e3 * t3, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "box-room map of bench.py, global map view, infer_bs = %d" % cfg.infer_bs,
                   "points": args.points, "neural_points": int(nm.neural_points.shape[0]),
                   "mask_fraction": float(mask3.mean().item())},
        "roofline": {"bound": "hbm", "kernel": "k_sdf_query_tile", "achieved": alg3 / t3 / 1e9, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": alg3 / t3 / 1e9 / HBM_PEAK_GBS, "traffic": pmc_traffic("k_sdf_query", args.points == 4194304),
                     "algorithmic_bytes_per_launch": alg3 * min(cfg.infer_bs, args.points) / args.points},
    }

    if not args.no_cpu_baseline:
        from oracle import cpu_ref as O

        threads = min(16, os.cpu_count() or 1)
        torch.set_num_threads(threads)
        cpu = lambda t: t.detach().cpu().clone()
        dx, mvd = O.search_neighborhood(cfg.num_nei_cells, cfg.search_alpha, cfg.voxel_size_m)
        st = O.MapState(
            buffer_pt_index=cpu(nm.buffer_pt_index), neural_points=cpu(nm.neural_points),
            point_ts_create=cpu(nm.point_ts_create), travel_dist=cpu(nm.travel_dist), cur_ts=int(nm.cur_ts),
            global2local=cpu(nm.global2local), local_neural_points=cpu(nm.local_neural_points),
            local_geo_features=cpu(nm.local_geo_features.data), local_point_certainties=cpu(nm.local_point_certainties),
            local_point_ts_update=cpu(nm.local_point_ts_update), resolution=cfg.voxel_size_m,
            buffer_size=cfg.buffer_size, diff_travel_dist_local=nm.diff_travel_dist_local, neighbor_dx=dx,
            max_valid_dist2=mvd, layer_norm_on=cfg.layer_norm_on, weighted_first=cfg.weighted_first,
        )
        od = O.DecoderParams(*[cpu(p) for p in dec.flat_params()], sdf_scale=dec.sdf_scale)
        xs = x[:262144].cpu()
        t0 = time.perf_counter()
        reps = 0
        while time.perf_counter() - t0 < 8.0:
            O.sdf_at(st, od, xs)
            reps += 1
        dt = time.perf_counter() - t0
        line3["cpu_baseline"] = {"value": xs.shape[0] * reps / dt, "unit": "points/s", "cores": threads, "kind": "port",
                                 "sample": f"{reps} x {xs.shape[0]} points through the CPU oracle's query+decode, {dt:.1f} s"}
        t0 = time.perf_counter()
        reps = 0
        while time.perf_counter() - t0 < 8.0:
            O.h_model(st, od, rot, sensor, pc_imu.cpu(), cfg.track_mask_query_nn_k, cfg.reg_min_grad_norm,
                      cfg.reg_max_grad_norm)
            reps += 1
        dt = time.perf_counter() - t0
        line1["cpu_baseline"] = {"value": n1 * reps / dt, "unit": "points/s", "cores": threads, "kind": "port",
                                 "sample": f"{reps} h_model evaluations of the CPU oracle on the same {n1} points, {dt:.1f} s"}
    line2 = sampler_row(cfg, args)
    print(json.dumps(line1))
    print(json.dumps(line2))
    print(json.dumps(line3))


if __name__ == "__main__":
    main()
