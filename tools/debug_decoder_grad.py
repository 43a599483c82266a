#!/usr/bin/env python3
"""Developer tool (GPU box): decoder-gradient error of one fused iteration against the CPU oracle, per component and batch size."""
import json
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import golden_io as gio  # noqa: E402
import shim_io as env  # noqa: E402
from oracle import cpu_ref as O  # noqa: E402
from test_hip_parity import _fused_grads  # noqa: E402
from test_tile_decode import _inputs  # noqa: E402
from clid_slam_amd import _lib  # noqa: E402

torch.set_num_threads(16)
H, D = _lib.H, _lib.D
for bs in (16384, 65536, 262144):
    for variant in (0, 1):
        p, g, cfg, index = _inputs(env, bs, seed=11)
        grad, loss, cert, ts = _fused_grads(env, cfg, p, g, index, split=True, variant=variant)
        st = gio.map_state()
        st.local_geo_features = gio.T(p["base_geo_features"])[gio.T(gio.load("state.npz")["local_mask"])].clone()
        pool, _ = gio.sample_pool()
        o = O.loss_and_grads(st, gio.decoder(g, "init_"), pool, index.to(torch.int64), O.LoopConfig())
        parts = {"W1": grad[: H * D], "b1": grad[H * D: H * D + H], "W2": grad[H * D + H: H * D + 2 * H], "b2": grad[H * D + 2 * H: H * D + 2 * H + 1]}
        gmax = max(float(o["grad_" + n].abs().max()) for n in parts)
        row = {"bs": bs, "variant": variant, "gmax": gmax}
        for n, t in parts.items():
            ref = o["grad_" + n].reshape(-1)
            d = (t - ref).abs()
            i = int(d.argmax())
            row[n] = {"max_abs": float(d.max()), "rel_to_gmax": float(d.max()) / gmax, "own_max": float(ref.abs().max()), "argmax": i,
                      "hip": float(t[i]), "ref": float(ref[i])}
        print(json.dumps(row))
