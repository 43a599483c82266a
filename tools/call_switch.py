"""Developer tool: cost of a mapping(K) call as a function of the calls before it (host enqueue | GPU events | wall, us).
   usage: python tools/call_switch.py "20 20 1 20 20 10 20 6 20 5 20"  """
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from clid_slam_amd import HotPathConfig
cfg = HotPathConfig(); cfg.device = "cuda:0"
nm, dec, mp, scene = bench.build_scene(cfg, "cuda:0")
mp.reserve(20)
for _ in range(50): mp.mapping(20)
torch.cuda.synchronize()
seq = [int(v) for v in sys.argv[1].split()]
out = []
for k in seq:
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record(); mp.mapping(k); e1.record(); th = time.perf_counter() - t0; torch.cuda.synchronize(); dt = time.perf_counter() - t0
    out.append((k, round(th * 1e6), round(e0.elapsed_time(e1) * 1e3), round(dt * 1e6)))
print(" ".join(f"[{k}: host {h} gpu {g} wall {w}]" for k, h, g, w in out))
