"""Developer tool: does replaying Mapper.mapping(K) as a captured HIP graph shorten the K-iteration chain?  (timing only: a
replay repeats the captured call's batches)   usage: python tools/graph_replay.py [K]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HSA_ENABLE_INTERRUPT", "0")
import torch, bench
from clid_slam_amd import HotPathConfig
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
cfg = HotPathConfig(); cfg.device = "cuda:0"
nm, dec, mp, scene = bench.build_scene(cfg, "cuda:0")
mp.reserve(K)
for _ in range(5): mp.mapping(K)
torch.cuda.synchronize()
def timed(fn, reps=30):
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter(); e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append((e0.elapsed_time(e1) * 1e3, (time.perf_counter() - t0) * 1e6))
    ts.sort(); return ts[len(ts) // 2]
print("eager   mapping(%d): gpu %.1f us  wall %.1f us" % ((K,) + timed(lambda: mp.mapping(K))))
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.stream(s):
        mp.mapping(K)  # warm on the side stream
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            mp.mapping(K)
    torch.cuda.synchronize()
    print("replay  mapping(%d): gpu %.1f us  wall %.1f us" % ((K,) + timed(lambda: g.replay())))
except Exception as e:
    print("capture failed:", type(e).__name__, str(e)[:300])
