"""`weighted_first: False` mapping calls on the bench scene: the fused iteration (csrc/train_wf0.hip, hoisted schedule)
against the un-fused loop over the autograd ops, ms per iteration.  python tools/time_wf0.py [iters]"""
import json
import sys
import time

import torch

sys.path.insert(0, ".")
import bench
from clid_slam_amd import HotPathConfig

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
out = {}
for name, pipeline in (("fused", 1), ("unfused", 0)):
    cfg = HotPathConfig()
    cfg.device = "cuda:0"
    cfg.weighted_first = False
    nm, dec, mp, scene = bench.build_scene(cfg, "cuda:0")
    mp.pipeline = pipeline
    gen = torch.Generator().manual_seed(5)
    idx = torch.randint(0, mp.pool_sample_count, (iters, cfg.bs), generator=gen).cuda()
    for _ in range(2):
        mp.mapping(iters, index_seq=idx)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        mp.mapping(iters, index_seq=idx)
    torch.cuda.synchronize()
    out[name + "_ms_per_iter"] = (time.perf_counter() - t0) / (reps * iters) * 1e3
    out[name + "_loss"] = float(mp.last_losses[-1, 0])
print(json.dumps(out))
