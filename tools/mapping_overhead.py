"""Developer tool: where the time of one Mapper.mapping(n) call goes (host enqueue vs GPU), on the bench scene."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from clid_slam_amd import HotPathConfig

cfg = HotPathConfig(); cfg.device = "cuda:0"
nm, dec, mp, scene = bench.build_scene(cfg, "cuda:0")
sync = torch.cuda.synchronize
for n in (10, 20, 200):
    mp.mapping(n); sync()
    ts = []
    for rep in range(30):
        sync(); t0 = time.perf_counter(); mp.mapping(n); t1 = time.perf_counter(); sync(); t2 = time.perf_counter()
        ts.append((t1 - t0, t2 - t0))
    ts.sort(key=lambda x: x[1])
    h, w = ts[len(ts) // 2]
    print(f"mapping({n}): host enqueue {h*1e6:.0f} us, wall {w*1e6:.0f} us = {w/n*1e6:.1f} us/iter (min wall {ts[0][1]*1e6:.0f})")
# pieces
def timeit(f, reps=50):
    f(); sync()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    t1 = time.perf_counter(); sync(); t2 = time.perf_counter()
    return (t1 - t0) / reps * 1e6, (t2 - t0) / reps * 1e6
print("draw_index(20):  host %.1f us, wall %.1f us" % timeit(lambda: mp._draw_index(20, 16384)))
print("5 x zeros:       host %.1f us, wall %.1f us" % timeit(lambda: [torch.zeros(200000, device='cuda') for _ in range(5)]))
print("assign_l2g:      host %.1f us, wall %.1f us" % timeit(lambda: nm.assign_local_to_global()))
print("_map_view:       host %.1f us, wall %.1f us" % timeit(lambda: nm._map_view(True)))
