"""Developer tool: host wall time (no extra synchronisation) inside the calls Mapper.process_frame / Mapper.mapping make, over the
steady-state frames of the sequence workload: where the host is when the device idles.  python tools/frame_host_stages.py [frames]"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench_sequence as BS
from clid_slam_amd import mapper as M, neural_points as NP, local_point_cloud_map as LP, tools as T, _lib

acc, order = {}, []
state = {"on": False}
def wrap(obj, name, label=None):
    fn = getattr(obj, name)
    label = label or name
    def w(*a, **k):
        if not state["on"]:
            return fn(*a, **k)
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            acc[label] = acc.get(label, 0.0) + time.perf_counter() - t0
    setattr(obj, name, w)
    order.append(label)

wrap(M.Mapper, "process_frame")
wrap(M.Mapper, "mapping")
wrap(LP.LocalPointCloudMap, "update_map")
wrap(M.Mapper, "_sample_compact_fused")
wrap(T, "voxel_down_sample_launch")
wrap(T, "voxel_down_sample_finish")
wrap(M.Mapper, "_pool_append_filter_fused")
wrap(NP.NeuralPoints, "update", "NeuralPoints.update")
wrap(NP.NeuralPoints, "_ensure_global_capacity")
wrap(NP.NeuralPoints, "_reset_local_map_fused")
wrap(M.Mapper, "determine_used_pose")
wrap(M.Mapper, "_new_sample_launch_pending")
wrap(NP.NeuralPoints, "prefetch_local_table")
wrap(NP.NeuralPoints, "_table")
wrap(M.Mapper, "_pool_filter_finish")
wrap(NP.NeuralPoints, "set_search_neighborhood")
wrap(M.Mapper, "_prepare_call")
wrap(NP.NeuralPoints, "_map_view")
wrap(NP.NeuralPoints, "assign_local_to_global")
wrap(_lib, "read_counts")
wrap(_lib, "small_to_host")

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 100
n = {"f": 0}
orig = M.Mapper.process_frame
def counted(self, *a, **k):
    n["f"] += 1
    state["on"] = n["f"] > frames // 2
    return orig(self, *a, **k)
M.Mapper.process_frame = counted
BS.run(frames, "cuda:0", quiet=True)
k = frames - frames // 2
print(json.dumps({lab: round(1e6 * acc.get(lab, 0.0) / k, 1) for lab in order}, indent=1))
