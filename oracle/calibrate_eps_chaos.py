#!/usr/bin/env python3
"""Calibration of the parameter-drift bounds of the config-5 (run_SubT_MRS.yaml) replay tests -- test infrastructure, CPU only.

The reference's Adam runs with eps = 1e-15 (utils/tools.py:205-255, config adam_eps), so the first step of ANY non-zero
gradient entry is +-lr whatever its magnitude.  On the SubT workload new neural points start with all-zero features
(feature_std 0) and pass through F.layer_norm (model/neural_points.py:632-633): rstd = 1/sqrt(1e-5) = 316 on a constant
row, and the row's gradient is 316 (dy - mean(dy)) -- differences of nearly equal sums whose sign depends on the summation
order.  Two CORRECT evaluations of the reference's own loop therefore disagree on a small fraction of the entries by up to
lr * iters.  This script measures that spread with the oracle against ITSELF: the same state, batches and code, once with
1 torch thread and once with 16 (different reduction orders inside index_add / matmul), for the state the replay tests
see (zero features + layer norm) and for two control states (random features; layer norm off), at several iteration counts.
It also evaluates the noise mask the tests use (entries of touched rows whose recorded gradient is below 1e-10 in some
iteration) -- how many of the disagreeing entries the mask explains.

    python oracle/calibrate_eps_chaos.py            # rewrites tests/golden/eps_chaos_calibration.json (~2 min)

tests/test_sequence.py and bench_sequence.py derive their bounds from that file (safety factor stated there)."""
from __future__ import annotations

import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402

import golden_io as gio  # noqa: E402
from oracle import cpu_ref as O  # noqa: E402


def run(threads: int, zero_feat: bool, ln: bool, iters: int, bs: int, frozen: bool):
    torch.set_num_threads(threads)
    st = gio.map_state(layer_norm_on=ln)
    p = gio.load("pool.npz")
    feat = gio.T(p["base_geo_features"])[gio.T(gio.load("state.npz")["local_mask"])].clone()
    st.local_geo_features = torch.zeros_like(feat) if zero_feat else feat
    dec = gio.decoder(gio.load("g6_loop_numerical_train_ln0.npz"), "init_")
    pool, _ = gio.sample_pool()
    gen = torch.Generator().manual_seed(1234)
    idx = torch.randint(0, pool.global_coord.shape[0], (iters, bs), generator=gen)
    lc = O.LoopConfig()
    lc.train_decoder = not frozen
    return O.mapping_iters(st, dec, pool, idx, lc, record=True)


def compare(a, b):
    """Per iteration count k (the state after k iterations of the same call): entries beyond 1e-4, the maximum, the decoder
    drift, and how many of the entries beyond 1e-4 lie outside the tests' noise mask."""
    rows = []
    noise = torch.zeros_like(a[0]["theta"], dtype=torch.bool)
    for k, (ra, rb) in enumerate(zip(a, b)):
        for r in (ra, rb):
            ga = r["grad_theta"].abs()
            noise |= (ga < 1e-10) & (ga.max(dim=1, keepdim=True).values > 0)
        d = (ra["theta"] - rb["theta"]).abs()
        big = d > 1e-4
        rows.append({
            "iters": k + 1, "entries": int(d.numel()), "n_gt_1e4": int(big.sum()), "frac_gt_1e4": float(big.float().mean()),
            "max": float(d.max()), "n_gt_1e4_outside_noise_mask": int((big & ~noise).sum()),
            "max_outside_noise_mask": float(d[~noise].max()) if (~noise).any() else 0.0,
            "decoder_max": max(float((x - y).abs().max()) for x, y in zip(ra["dec"], rb["dec"])),
            "loss_diff": abs(float(ra["loss"]) - float(rb["loss"])),
            "noise_mask_entries": int(noise.sum()),
        })
    return rows


def main():
    iters, bs = 10, 16384
    out = {"what": "oracle (CPU restatement of utils/mapper.py:620-862) against itself, 1 vs 16 torch threads, same state / batches; "
                   "tests/golden map state (M_local rows x 8 features), bs 16384, numerical eikonal, Adam eps 1e-15",
           "torch": torch.__version__, "cases": {}}
    for name, zero, ln, frozen in (("zero_features_layer_norm", True, True, False),
                                   ("zero_features_layer_norm_frozen_decoder", True, True, True),
                                   ("random_features_layer_norm", False, True, False),
                                   ("zero_features_no_layer_norm", True, False, False),
                                   ("random_features_no_layer_norm", False, False, False)):
        a = run(1, zero, ln, iters, bs, frozen)
        b = run(16, zero, ln, iters, bs, frozen)
        out["cases"][name] = compare(a, b)
        last = out["cases"][name][-1]
        print(f"{name:44s} after {iters} iterations: {last['n_gt_1e4']:5d} of {last['entries']} entries > 1e-4 (max {last['max']:.2e}), "
              f"{last['n_gt_1e4_outside_noise_mask']} outside the noise mask, decoder {last['decoder_max']:.2e}", file=sys.stderr)
    path = os.path.join(ROOT, "tests", "golden", "eps_chaos_calibration.json")
    with open(path, "w") as fh:
        json.dump(out, fh, indent=1)
    print("wrote", path, file=sys.stderr)


if __name__ == "__main__":
    main()
