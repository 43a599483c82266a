#!/usr/bin/env python3
"""Developer tool (GPU box): which N2 label / mask mismatches between the HIP sampler kernels and the reference's G9 outputs
does oracle.sampler_ref.region_sdf_ambiguity name?  Prints, per frame, the mismatching rows with their margins."""
import json
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import golden_io as gio  # noqa: E402
from oracle import sampler_ref as R  # noqa: E402
from test_sampler_gpu import _cfg, _cloud_at  # noqa: E402

g = gio.load("g9_sampler.npz")
for fid in (0, 1, 2):
    cfg = _cfg(g)
    lpm = _cloud_at(g, fid, cfg)
    q = gio.T(g[f"f{fid}_q"])
    d, ok = lpm.region_specific_sdf_estimation(q.cuda())
    d, ok = d.cpu(), ok.cpu()
    lc = R.LocalCloud.empty(resolution=0.2, buffer_size=cfg.local_buffer_size, map_size=cfg.local_map_size)
    lc.buffer_pt_index, lc.points = lpm.buffer_pt_index.cpu(), lpm.local_point_cloud_map.cpu()
    amb = R.region_sdf_ambiguity(lc, q)
    err = (d - gio.T(g[f"f{fid}_q_sdf"])).abs()
    bad = err > 1e-4
    rows = []
    for i in torch.nonzero(bad).flatten().tolist():
        rows.append({"i": i, "err": float(err[i]), **{k: bool(v[i]) for k, v in amb.items()}})
    print(json.dumps({"frame": fid, "n": int(q.shape[0]), "bad": int(bad.sum()), "bad_named": int((bad & (amb["any"] & ~amb["cell"])).sum()),
                      "named": {k: int(v.sum()) for k, v in amb.items()}, "max_err_unnamed": float(err[~(amb["any"])].max()),
                      "p99.9_err_unnamed": float(np.quantile(err[~amb["any"]].numpy(), 0.999)), "rows": rows[:20]}))
