"""developer tool (GPU box): hit-count histogram of the cell-directory search on the bench scene and the first record that
differs from the probing kernels."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import bench, test_celldir_gpu as T
from clid_slam_amd import HotPathConfig

cfg = HotPathConfig(); cfg.device = "cuda:0"
nm, dec, mp, scene = bench.build_scene(cfg, "cuda:0")
view, keep, d = T._directory(nm)
print("box", d["o"], d["n"], "words", d["words"], "hits", d["n_hits"], "M", nm.local_count())
S = mp.pool_sample_count
g = torch.Generator().manual_seed(1)
q = mp.global_coord_pool[torch.randint(0, S, (20000,), generator=g).cuda()].cpu().numpy()
cell = np.floor(q / np.float32(cfg.voxel_size_m)).astype(np.int64)
dx = nm.neighbor_dx.cpu().numpy().astype(np.int64)
H = np.zeros(len(q), dtype=np.int64)
for o in range(dx.shape[0]):
    H += (T._chain(nm, cell + dx[o], True) >= 0)
print("H percentiles", np.percentile(H, [10, 50, 90, 99, 100]), "frac H>32", (H > 32).mean(), "frac H>24", (H > 24).mean(), "mean", H.mean())
for bs, iters in ((16384, 1),):
    idx = torch.randint(0, S, (iters, bs), device="cuda", generator=torch.Generator("cuda").manual_seed(3 + bs))
    a = T._records(nm, mp, idx, bs, cfg.gradient_decimation, 0)
    b = T._records(nm, mp, idx, bs, cfg.gradient_decimation, 8)
    ra, rb = a[0].contiguous().view(torch.int32), b[0].contiguous().view(torch.int32)
    bad = torch.nonzero((ra != rb).any(-1).any(-1)).flatten()
    print("tasks differing", bad.numel(), "of", ra.shape[0])
    for t in bad[:3].tolist():
        fa, fb = a[0][t], b[0][t]
        for sl in range(8):
            wa, wb = fa[16 + 4 * sl:20 + 4 * sl].reshape(8, 2), fb[16 + 4 * sl:20 + 4 * sl].reshape(8, 2)
            if not torch.equal(wa.view(torch.int32), wb.view(torch.int32)):
                x = fa[sl, :3].numpy()
                c = np.floor(x / np.float32(cfg.voxel_size_m)).astype(np.int64)
                h = sum(int(T._chain(nm, (c + dx[o])[None], True)[0] >= 0) for o in range(dx.shape[0]))
                print(" task", t, "slot", sl, "x", x, "H", h, "\n   dir ", wa[:6, 0].tolist(), wa[:6, 1].contiguous().view(torch.int32).tolist(),
                      "\n   prob", wb[:6, 0].tolist(), wb[:6, 1].contiguous().view(torch.int32).tolist())
