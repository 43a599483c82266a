"""developer tool (GPU box): on the sequence workload's state after frame k, compare the directory search records with probing."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import bench_sequence as BS, test_celldir_gpu as T

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 2
cfg, rows, checks, (nm, dec, mp) = BS.run(frames, "cuda:0", quiet=True)
view, keep, d = T._directory(nm)
print("box", d["o"], d["n"], "nzw", d["nzw"], "words", d["words"], "valid", d["valid"], "hits", d["n_hits"], "M", nm.local_count(), "log2filter", view.log2filter)
bs = 16384
idx = torch.randint(0, mp.pool_sample_count, (2, bs), device="cuda", generator=torch.Generator("cuda").manual_seed(3))
a = T._records(nm, mp, idx, bs, cfg.gradient_decimation, 0)
b = T._records(nm, mp, idx, bs, cfg.gradient_decimation, 8)
ra, rb = a[0].contiguous().view(torch.int32), b[0].contiguous().view(torch.int32)
bad = torch.nonzero((ra != rb).any(-1).any(-1)).flatten()
print("tasks differing", bad.numel(), "of", ra.shape[0], "deferred", a[3])
dx = nm.neighbor_dx.cpu().numpy().astype(np.int64)
for t in bad[:4].tolist():
    fa, fb = a[0][t], b[0][t]
    for sl in range(8):
        wa, wb = fa[16 + 4 * sl:20 + 4 * sl].reshape(8, 2), fb[16 + 4 * sl:20 + 4 * sl].reshape(8, 2)
        if not torch.equal(wa.view(torch.int32), wb.view(torch.int32)):
            x = fa[sl, :3].numpy()
            c = np.floor(x / np.float32(cfg.voxel_size_m)).astype(np.int64)
            ch = [int(T._chain(nm, (c + dx[o])[None], True)[0]) for o in range(dx.shape[0])]
            print(" task", t, "slot", sl, "x", x, "cell", c, "H", sum(v >= 0 for v in ch), "\n   dir ", wa[:6, 0].tolist(), wa[:6, 1].contiguous().view(torch.int32).tolist(),
                  "\n   prob", wb[:6, 0].tolist(), wb[:6, 1].contiguous().view(torch.int32).tolist())
            break
# which query points fall outside the box (they defer their task to the probing launch)?
S = mp.pool_sample_count
q = mp.global_coord_pool[torch.randint(0, S, (200000,), device="cuda")].cpu().numpy()
cell = np.floor(q / np.float32(cfg.voxel_size_m)).astype(np.int64)
o, n = np.array(d["o"]), np.array(d["n"])
rel = cell - o
for margin_extra in (0, 4, 8, 12):
    lo = 2 - margin_extra
    out = (rel < lo) | (rel >= n - 2 + margin_extra)
    print("extra margin", margin_extra, "fraction of samples outside", out.any(1).mean(), "per axis", out.mean(0))
