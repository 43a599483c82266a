"""Developer tool: wall-clock breakdown of the host side of Mapper.mapping(10) (wrapped sub-calls, no cProfile)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from clid_slam_amd import HotPathConfig, _lib, mapper as MP, neural_points as NP
cfg = HotPathConfig(); cfg.device = "cuda:0"
nm, dec, mp, scene = bench.build_scene(cfg, "cuda:0")
mp.reserve(10)
acc = {}
def wrap(obj, name, label=None):
    f = getattr(obj, name); label = label or name
    def g(*a, **k):
        t = time.perf_counter_ns(); r = f(*a, **k); acc[label] = acc.get(label, 0) + time.perf_counter_ns() - t; return r
    setattr(obj, name, g)
lib = _lib.load()
for n in ("_loop_buffers", "_draw_index", "_check_fused_config"): wrap(mp, n)
wrap(nm, "_map_view"); wrap(nm, "assign_local_to_global"); wrap(dec, "flat_params")
wrap(_lib, "stream"); wrap(_lib, "require_cuda"); wrap(MP, "_dist")
class L:  # proxy for the C call
    def __getattr__(self, k): return getattr(lib, k)
    def clid_mapping_run(self, *a):
        t = time.perf_counter_ns(); r = lib.clid_mapping_run(*a); acc["C clid_mapping_run"] = acc.get("C clid_mapping_run", 0) + time.perf_counter_ns() - t; return r
_lib_load = _lib.load
_lib.load = lambda: L()
for _ in range(20): mp.mapping(10); torch.cuda.synchronize()
acc.clear(); tot = 0
N = 200
for _ in range(N):
    t = time.perf_counter_ns(); mp.mapping(10); tot += time.perf_counter_ns() - t; torch.cuda.synchronize()
print("mapping() host total us", tot / N / 1e3)
for k, v in sorted(acc.items(), key=lambda x: -x[1]): print("  %-26s %7.2f us" % (k, v / N / 1e3))
