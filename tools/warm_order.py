"""developer tool (GPU box): does the ORDER of bench.py's set-up change the first timed call?  A: build_scene, reserve, gc.collect,
5 x (mapping(1), sync), timed mapping(20) [bench.py]; B: gc.collect first, then build_scene, reserve, warm-up, timed.
usage: python tools/warm_order.py A|B"""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HSA_ENABLE_INTERRUPT", "0")
import torch, bench
from clid_slam_amd import HotPathConfig
mode = sys.argv[1]
cfg = HotPathConfig(); cfg.device = "cuda:0"
if mode == "B":
    gc.collect(); gc.disable()
nm, dec, mp, scene = bench.build_scene(cfg, "cuda:0")
mp.reserve(20)
if mode == "A":
    gc.collect(); gc.disable()
for _ in range(5):
    mp.mapping(1); torch.cuda.synchronize()
torch.cuda.synchronize()
t0 = time.perf_counter(); mp.mapping(20); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(mode, "first timed mapping(20):", round(dt * 1e6, 1), "us")
