"""developer tool (GPU box): how the end of bench.py's timed region is detected -- event poll then synchronize (the bench's way) vs
synchronize alone -- wall clock of a mapping(20) call, alternating, medians.  usage: python tools/region_end_timing.py"""
import json, os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HSA_ENABLE_INTERRUPT", "0")
import torch, bench, gc
from clid_slam_amd import HotPathConfig
cfg = HotPathConfig(); cfg.device = "cuda:0"
nm, dec, mp, scene = bench.build_scene(cfg, "cuda:0")
mp.reserve(20)
for _ in range(5): mp.mapping(20)
torch.cuda.synchronize(); gc.collect(); gc.disable()
res = {"poll_then_sync": [], "sync_only": [], "poll_only": []}
for rep in range(30):
    for mode in res:
        mp.mapping(1); torch.cuda.synchronize()
        ev1 = torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        mp.mapping(20)
        if mode != "sync_only":
            ev1.record()
            while not ev1.query():
                pass
        if mode != "poll_only":
            torch.cuda.synchronize()
        res[mode].append(time.perf_counter() - t0)
print(json.dumps({k: round(1e3 * statistics.median(v), 4) for k, v in res.items()}), "(ms per mapping(20) call, median of 30)")
