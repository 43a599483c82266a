"""Developer tool: host time between the landmarks of process_frame's tail (after the insert / window read-back), steady state.
python tools/frame_tail_gaps.py [frames]"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench_sequence as BS
from clid_slam_amd import mapper as M, neural_points as NP, _lib

marks, acc, cnt = [], {}, [0]
on = {"v": False}
def mark(name):
    if on["v"]:
        marks.append((name, time.perf_counter()))
def wrap(obj, name, label=None):
    fn = getattr(obj, name); label = label or name
    def w(*a, **k):
        mark(label + ":in")
        try:
            return fn(*a, **k)
        finally:
            mark(label + ":out")
    setattr(obj, name, w)
wrap(_lib, "read_counts")
wrap(NP.NeuralPoints, "_reset_local_map_fused", "window")
wrap(NP.NeuralPoints, "update", "update")
wrap(NP.NeuralPoints, "prefetch_local_table", "prefetch")
wrap(NP.NeuralPoints, "_table", "_table")
wrap(M.Mapper, "_new_sample_launch_pending", "newsel")
wrap(M.Mapper, "_pool_filter_finish", "poolfinish")
wrap(M.Mapper, "determine_used_pose", "usedpose")
lib = _lib.load()
orig_tb = lib.clid_table_build
class L:
    def __getattr__(self, k): return getattr(lib, k)
    def clid_table_build(self, *a):
        mark("clid_table_build:in"); r = orig_tb(*a); mark("clid_table_build:out"); return r
_lib.load = lambda: L()
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 100
orig = M.Mapper.process_frame
def pf(self, *a, **k):
    cnt[0] += 1
    on["v"] = cnt[0] > frames // 2
    marks.clear()
    mark("frame:in")
    r = orig(self, *a, **k)
    mark("frame:out")
    if on["v"]:
        seen = {}
        for (n0, t0), (n1, t1) in zip(marks[:-1], marks[1:]):
            key = n0 + " -> " + n1
            seen[key] = seen.get(key, 0) + 1
            key += " #%d" % seen[key]
            acc.setdefault(key, []).append(t1 - t0)
    return r
M.Mapper.process_frame = pf
BS.run(frames, "cuda:0", quiet=True)
for k, v in acc.items():
    if len(v) > 10:
        print("%-48s %7.1f us  (n=%d)" % (k, 1e6 * sum(v) / len(v), len(v)))
