"""Where one Mapper.process_frame + mapping() frame spends its time on the GPU box (developer tool).
Wraps the stages with synchronised wall-clock timers; run a few frames of a moving sensor."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from clid_slam_amd import DataSampler, Decoder, HotPathConfig, LocalPointCloudMap, Mapper, NeuralPoints
from clid_slam_amd.synth import box_room_scan
from clid_slam_amd import tools as T

cfg = HotPathConfig(); cfg.device = "cuda:0"
dev = "cuda:0"
acc = {}
def wrap(obj, name, label=None):
    fn = getattr(obj, name)
    label = label or name
    def w(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = fn(*a, **k)
        torch.cuda.synchronize(); acc.setdefault(label, []).append(time.perf_counter() - t0)
        return r
    setattr(obj, name, w)

class DS:
    lose_track = False; stop_status = False; processed_frame = 0; gt_pose_provided = False

nm = NeuralPoints(cfg); nm.travel_dist = torch.arange(16, device=dev) * 0.5
lpm = LocalPointCloudMap(cfg)
mp = Mapper(cfg, DS(), nm, lpm, Decoder(cfg, cfg.geo_mlp_hidden_dim, cfg.geo_mlp_level, 1))
mp.sampler = DataSampler(cfg)
wrap(lpm, "update_map", "raw-point map update"); wrap(mp.sampler, "sample", "sampler (launch + compaction)")
wrap(nm, "update", "NeuralPoints.update (+reset_local_map)"); wrap(nm, "query_certainty", "query_certainty")
wrap(mp, "process_frame", "process_frame TOTAL"); wrap(mp, "mapping", "mapping(10) TOTAL")
wrap(nm, "assign_local_to_global", "assign_local_to_global"); wrap(nm, "reset_local_map", "reset_local_map")
for fid in range(8):
    s = (0.5 * fid, 0.1 * fid, 1.5)
    scan = box_room_scan(sensor=s, seed=100 + fid).to(dev)
    pose = torch.eye(4, dtype=torch.float64, device=dev); pose[:3, 3] = torch.tensor(s, dtype=torch.float64)
    DS.processed_frame = fid
    mp.process_frame(scan, None, pose, fid)
    mp.mapping(10)
print("pool", mp.pool_sample_count, "points", nm.count(), "local", nm.local_count())
for k, v in acc.items():
    v = v[2:] if len(v) > 3 else v
    print(f"{k:42s} n={len(v):3d} mean {1e3 * sum(v) / len(v):8.3f} ms")

if os.environ.get("CLID_TORCHPROF"):
    from torch.profiler import profile, ProfilerActivity
    fid = 9
    s = (0.5 * fid, 0.1 * fid, 1.5)
    scan = box_room_scan(sensor=s, seed=100 + fid).to(dev)
    pose = torch.eye(4, dtype=torch.float64, device=dev); pose[:3, 3] = torch.tensor(s, dtype=torch.float64)
    DS.processed_frame = fid
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        mp.process_frame(scan, None, pose, fid)
        torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=14, max_name_column_width=50))
    ops = sorted(((e.count, e.key) for e in prof.key_averages() if e.key.startswith("aten::")), reverse=True)[:28]
    print("aten op counts:", ", ".join(f"{k[6:]}x{c}" for c, k in ops))
    print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=14, max_name_column_width=50))

if os.environ.get("CLID_CPROFILE"):
    import cProfile, pstats
    fid = 8
    s = (0.5 * fid, 0.1 * fid, 1.5)
    scan = box_room_scan(sensor=s, seed=100 + fid).to(dev)
    pose = torch.eye(4, dtype=torch.float64, device=dev); pose[:3, 3] = torch.tensor(s, dtype=torch.float64)
    DS.processed_frame = fid
    pr = cProfile.Profile(); pr.enable()
    mp.process_frame(scan, None, pose, fid)
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(18)
