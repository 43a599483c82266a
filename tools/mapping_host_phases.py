"""Developer tool (GPU box): where the host time of one Mapper.mapping(20) call goes -- entry -> clid_mapping_prep -> argument
assembly -> clid_mapping_run -> write-back -> return (microseconds, mean of 200 calls)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from clid_slam_amd import HotPathConfig, _lib
cfg = HotPathConfig(); cfg.device = "cuda:0"
nm, dec, mp, scene = bench.build_scene(cfg, "cuda:0")
mp.reserve(20)
lib = _lib.load()
marks = []
class L:
    def __getattr__(self, k): return getattr(lib, k)
    def clid_mapping_prep(self, *a):
        marks.append(("prep_start", time.perf_counter_ns())); r = lib.clid_mapping_prep(*a); marks.append(("prep_end", time.perf_counter_ns())); return r
    def clid_mapping_run(self, *a):
        marks.append(("run_start", time.perf_counter_ns())); r = lib.clid_mapping_run(*a); marks.append(("run_end", time.perf_counter_ns())); return r
_lib.load = lambda: L()
for _ in range(20): mp.mapping(20); torch.cuda.synchronize()
acc = {}
N = 200
for _ in range(N):
    marks.clear()
    t0 = time.perf_counter_ns(); mp.mapping(20); t1 = time.perf_counter_ns(); torch.cuda.synchronize()
    m = dict(marks)
    for k, v in (("entry -> prep", m["prep_start"] - t0), ("clid_mapping_prep (2 launches)", m["prep_end"] - m["prep_start"]),
                 ("argument assembly", m["run_start"] - m["prep_end"]), ("clid_mapping_run (enqueue of the loop)", m["run_end"] - m["run_start"]),
                 ("write-back + return", t1 - m["run_end"]), ("total", t1 - t0)):
        acc[k] = acc.get(k, 0) + v
for k, v in acc.items(): print("%-44s %7.2f us" % (k, v / N / 1e3))
