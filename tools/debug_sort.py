"""Developer tool (GPU box): clid_mapping_prep's ordered batches against the host restatement, narrow / wide composites, per decimation."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from clid_slam_amd import _lib
lib = _lib.load()
dev = "cuda:0"
pool, seed = 897_123, 42
torch.manual_seed(0)
coords = (torch.rand((pool, 3), device=dev) * 80.0 - 40.0).contiguous()
coords[: pool // 2] = torch.floor(coords[: pool // 2] / 3.2) * 3.2 + 0.1
iters2, bs2 = 2, 2 * 16384 + 1000
raw = torch.empty((iters2, bs2), device=dev, dtype=torch.int64)
_lib.check(lib.clid_mapping_prep(None, 0, raw.data_ptr(), iters2, bs2, 0, pool, None, 0, seed, 77, None, 0.4, None, 0, 0, 1, _lib.stream()), "prep")
ws2 = torch.empty(int(lib.clid_mapping_prep_workspace_bytes(iters2, bs2)), device=dev, dtype=torch.uint8)
raw_n, c2 = raw.cpu().numpy(), coords.cpu().numpy()
def morton(c):
    out = np.zeros(c.shape[:-1], dtype=np.int64)
    for b in range(8):
        for a in range(3):
            out |= ((c[..., a] >> b) & 1) << (3 * b + a)
    return out
def class_order(a, ka, col0, d):
    cols = col0 + np.arange(a.shape[0]); cls = cols % d == 0; out = np.empty_like(a)
    for c in (True, False):
        sel = np.nonzero(cls == c)[0]; out[sel] = a[sel][np.argsort(ka[sel], kind="stable")]
    return out
for wide in ("0", "1"):
    os.environ["CLID_SORT_WIDE"] = wide
    for d in (1, 10, 7, 3):
        srt = torch.empty_like(raw)
        _lib.check(lib.clid_mapping_prep(None, 0, srt.data_ptr(), iters2, bs2, 0, pool, None, 0, seed, 77, coords.data_ptr(), 0.4, ws2.data_ptr(), 0, 0, d, _lib.stream()), "prep")
        sd = srt.cpu().numpy()
        for it in range(iters2):
            for lo in range(0, bs2, 16384):
                a = raw_n[it, lo:lo + 16384]
                ka = morton(np.floor(c2[a] / np.float32(0.4)).astype(np.int64) & 255)
                want = class_order(a, ka, lo, d)
                got = sd[it, lo:lo + 16384]
                bad = np.nonzero(got != want)[0]
                perm = np.array_equal(np.sort(got), np.sort(a))
                print(f"wide={wide} d={d} it={it} lo={lo}: mismatches {len(bad)} perm={perm}" + (f" first {bad[:6]} last {bad[-3:]}" if len(bad) else ""))
