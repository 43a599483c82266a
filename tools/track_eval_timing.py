"""developer tool (GPU box): one tracking evaluation (tracking.normal_equations) on the sequence workload's map -- wall per evaluation,
enqueue-only host time, device time of the model kernel -- for scans of N points.  usage: python tools/track_eval_timing.py"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HSA_ENABLE_INTERRUPT", "0")
import torch
import bench_sequence as BS
from clid_slam_amd import tracking
from clid_slam_amd.tools import voxel_down_sample_torch
from clid_slam_amd.synth import hall_scan, sweep_poses
cfg, rows, checks, (nm, dec, mp) = BS.run(30, "cuda:0", quiet=True)
poses = sweep_poses(31)
pts = hall_scan(poses[29], seed=1029, device="cuda:0", min_range=cfg.min_range, max_range=cfg.max_range, vox_down_m=cfg.vox_down_m)
pose = poses[29].to("cuda:0")
rot, pos = pose[:3, :3].float().contiguous(), pose[:3, 3].float().contiguous()
for vox in (0.6, 1.0, 1.5):
    src = pts[voxel_down_sample_torch(pts, vox)].contiguous()
    for _ in range(20): tracking.normal_equations(nm, dec, cfg, rot, pos, src, host=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200): tracking.normal_equations(nm, dec, cfg, rot, pos, src, host=True)
    wall = (time.perf_counter() - t0) / 200
    b = tracking.bind(nm, dec, cfg, src)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200): b.launch(rot, pos, False, True, result=True)
    enq = (time.perf_counter() - t0) / 200
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): b.launch(rot, pos, False, True, result=True)
    e1.record(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2000): tracking.bind(nm, dec, cfg, src)
    tb = (time.perf_counter() - t0) / 2000
    print(json.dumps({"points": int(src.shape[0]), "vox_m": vox, "M_local": nm.local_count(), "wall_us_per_evaluation": round(1e6 * wall, 1),
                      "enqueue_only_us": round(1e6 * enq, 1), "device_us_back_to_back": round(1e3 * e0.elapsed_time(e1) / 200, 1),
                      "bind_lookup_us": round(1e6 * tb, 1)}))
