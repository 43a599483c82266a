"""Developer tool: how the first timed mapping(20) call depends on what ran before it (GPU clock ramp).
   usage: python tools/warm_clocks.py one|two|same|sleep|five"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from clid_slam_amd import HotPathConfig
cfg = HotPathConfig(); cfg.device = "cuda:0"
nm, dec, mp, scene = bench.build_scene(cfg, "cuda:0")
mp.reserve(20)
mode = sys.argv[1]
if mode == "one": mp.mapping(5)
elif mode == "two": mp.mapping(5); mp.mapping(5)
elif mode == "same": mp.mapping(20)
elif mode == "sleep": mp.mapping(5); torch.cuda.synchronize(); time.sleep(0.5)
elif mode == "five":
    for _ in range(5): mp.mapping(1)
elif mode == "fivesync":
    for _ in range(5): mp.mapping(1); torch.cuda.synchronize()
elif mode == "fivehot":  # the loop itself first (everything resident), then the driver's warm-up shape
    for _ in range(50): mp.mapping(20)
    torch.cuda.synchronize()
    for _ in range(5): mp.mapping(1)
elif mode == "fivehotsync":
    for _ in range(50): mp.mapping(20)
    torch.cuda.synchronize()
    for _ in range(5): mp.mapping(1); torch.cuda.synchronize()
elif mode == "busy":  # ~100 ms of matrix products, then the driver's warm-up shape
    a = torch.randn(4096, 4096, device="cuda:0")
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.1:
        a @ a
    torch.cuda.synchronize()
    for _ in range(5): mp.mapping(1)
elif mode == "loop":  # ~100 ms of the loop itself
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.1:
        mp.mapping(20)
    torch.cuda.synchronize()
    for _ in range(5): mp.mapping(1)
torch.cuda.synchronize()
for r in range(4):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record(); mp.mapping(20); e1.record(); th = time.perf_counter() - t0; torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(mode, r, "total", round(dt * 1e6, 1), "host-enqueue", round(th * 1e6, 1), "gpu", round(e0.elapsed_time(e1) * 1e3, 1), flush=True)
