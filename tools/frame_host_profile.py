"""Developer tool: cProfile of Mapper.process_frame over the steady-state frames of the sequence workload."""
import cProfile, os, pstats, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench_sequence as BS
from clid_slam_amd import mapper as M

pr = cProfile.Profile()
orig = M.Mapper.process_frame
state = {"n": 0}
def wrapped(self, *a, **k):
    state["n"] += 1
    if state["n"] > 25:
        pr.enable()
        try:
            return orig(self, *a, **k)
        finally:
            pr.disable()
    return orig(self, *a, **k)
M.Mapper.process_frame = wrapped
BS.run(65, "cuda:0", quiet=True)
st = pstats.Stats(pr)
print("frames profiled:", state["n"] - 25)
st.sort_stats("cumulative").print_stats(45)
