"""developer tool (GPU box, under rocprofv3 --kernel-trace --stats): rebuild the local table + cell directory of a 230 k-point map."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from clid_slam_amd import HotPathConfig, NeuralPoints, _lib
if os.environ.get("CLID_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["CLID_LIB"])
cfg = HotPathConfig(); cfg.device = "cuda:0"; cfg.buffer_size = 50_000_000
torch.manual_seed(11)
nm = NeuralPoints(cfg); nm.local_map_radius = 500.0; nm.travel_dist = torch.zeros(4, device="cuda:0")
g = torch.Generator().manual_seed(5)
u = torch.arange(240, dtype=torch.float32) * 0.4 - 48.0
uu, vv = torch.meshgrid(u, u, indexing="ij")
planes = []
for k, z in enumerate((0.1, 3.3, 6.5, 9.7)):
    jit = (torch.rand((uu.numel(), 3), generator=g) - 0.5) * 0.2
    planes.append(torch.stack((uu.reshape(-1) + 0.2, vv.reshape(-1) + 0.2, torch.full((uu.numel(),), z)), 1) + jit)
nm.update(torch.cat(planes).to("cuda:0"), torch.zeros(3, device="cuda:0"), torch.eye(3, device="cuda:0"), 0)
print("M", nm.local_count())
for i in range(20):
    nm._map_version += 1
    nm._map_view(True)
torch.cuda.synchronize()
print(nm._tables[(True, True)][4][0].tolist())
