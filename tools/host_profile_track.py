"""developer tool (GPU box): where the host time of one tracking-model call goes."""
import sys, os, cProfile, pstats, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from clid_slam_amd import HotPathConfig, tracking
cfg = HotPathConfig(); cfg.device = "cuda:0"
nm, dec, mp, scene = bench.build_scene(cfg, "cuda:0")
near = scene["sdf_label"].abs() < 0.02
surf = scene["coord"][near]
sensor = scene["sensor"].to(torch.float32)
pc = (surf[:8192] - sensor).cuda().contiguous()
rot, pos = torch.eye(3).cuda(), sensor.cuda()
for _ in range(100): tracking._launch(nm, dec, cfg, rot, pos, pc, False, True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(2000): tracking._launch(nm, dec, cfg, rot, pos, pc, False, True)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("host us per call", 1e6 * (t1 - t0) / 2000, "incl. drain", 1e6 * (t2 - t0) / 2000)
pr = cProfile.Profile(); pr.enable()
for _ in range(2000): tracking._launch(nm, dec, cfg, rot, pos, pc, False, True)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
