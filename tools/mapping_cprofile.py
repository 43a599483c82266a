"""Developer tool (GPU box): cProfile of the host side of Mapper.mapping(20) (2000 calls, device kept busy but never waited on
inside the profile: the enqueue cost is what shows)."""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from clid_slam_amd import HotPathConfig
cfg = HotPathConfig(); cfg.device = "cuda:0"
nm, dec, mp, scene = bench.build_scene(cfg, "cuda:0")
mp.reserve(20)
for _ in range(20): mp.mapping(20); torch.cuda.synchronize()
pr = cProfile.Profile()
def run():
    for _ in range(300):
        mp.mapping(20)
        torch.cuda.synchronize()
pr.enable(); run(); pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(28)
