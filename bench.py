#!/usr/bin/env python3
"""Headline benchmark: sampled points / second through one mapping iteration of CLID-SLAM's
SDF training loop (BASELINE.json `metric`), config `run_ncd128.yaml` defaults (configs[1]): fp32,
bs = 16384 per GPU, numerical eikonal term on every 10th sample, Adam step included.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one iteration of `Mapper.mapping` (get_batch gathers, query of bs + 6*ceil(bs/10)
points, decode, loss, backward, Adam).  N > 1: weak scaling -- every rank trains on its own 16384-
sample slice of a global batch of N*16384, gradients all-reduced over RCCL each iteration.
Prints ONE JSON line on rank 0.  Inputs are synthetic (box-room scan, seeds fixed) and resident in
HBM before the timed region.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X spec peak (MI355X_MICROARCH.md); 6290 GB/s is the measured copy ceiling
# algorithmic bytes per query point (restated in DESIGN.md)
BYTES_FWD_PER_QUERY = 1004.0
BYTES_BWD_PER_QUERY = 436.0
BYTES_POOL_GATHER = 24.0
BYTES_ADAM_PER_ROW = 256.0


def build_scene(cfg, device):
    """Synthetic scan -> sample pool + neural-point map through the product's own classes."""
    from clid_slam_amd import Decoder, Mapper, NeuralPoints
    from clid_slam_amd.synth import box_room_pool

    class _DS:
        lose_track = False
        stop_status = False
        processed_frame = 0
        gt_pose_provided = False

    d = box_room_pool(cfg)
    nm = NeuralPoints(cfg)
    nm.travel_dist = torch.zeros(4, device=device)
    near = d["sdf_label"].abs() < cfg.surface_sample_range_m * 0.5
    nm.update(d["coord"][near].to(device), d["sensor"].to(device), torch.eye(3, device=device), 0)
    gen = torch.Generator().manual_seed(42)
    nm.geo_features = (0.3 * torch.randn(nm.geo_features.shape, generator=gen)).to(device)
    nm.reset_local_map(d["sensor"].to(device), torch.eye(3, device=device), 0, reboot_map=True)
    torch.manual_seed(42)
    dec = Decoder(cfg, cfg.geo_mlp_hidden_dim, cfg.geo_mlp_level, 1)
    mp = Mapper(cfg, _DS(), nm, None, dec)
    S = d["coord"].shape[0]
    mp.set_pool(d["coord"], d["sdf_label"], d["weight"], torch.zeros(S, dtype=torch.int32))
    return nm, dec, mp, d


def cpu_baseline(cfg, nm, dec, mp, budget_s=15.0, max_iters=400):
    """The CPU oracle (port of the reference's PyTorch path) on the same map/pool, host cores."""
    from oracle import cpu_ref as O

    threads = min(16, os.cpu_count() or 1)  # slam.py:44 uses 16
    torch.set_num_threads(threads)
    cpu = lambda t: t.detach().cpu().clone()
    dx, mv = O.search_neighborhood(cfg.num_nei_cells, cfg.search_alpha, cfg.voxel_size_m)
    st = O.MapState(
        buffer_pt_index=cpu(nm.buffer_pt_index), neural_points=cpu(nm.neural_points),
        point_ts_create=cpu(nm.point_ts_create), travel_dist=cpu(nm.travel_dist), cur_ts=int(nm.cur_ts),
        global2local=cpu(nm.global2local), local_neural_points=cpu(nm.local_neural_points),
        local_geo_features=cpu(nm.local_geo_features.data), local_point_certainties=cpu(nm.local_point_certainties),
        local_point_ts_update=cpu(nm.local_point_ts_update), resolution=cfg.voxel_size_m, buffer_size=cfg.buffer_size,
        diff_travel_dist_local=nm.diff_travel_dist_local, neighbor_dx=dx, max_valid_dist2=mv,
        layer_norm_on=cfg.layer_norm_on, weighted_first=cfg.weighted_first,
    )
    od = O.DecoderParams(*[cpu(p) for p in dec.flat_params()], sdf_scale=dec.sdf_scale)
    pool = O.SamplePool(cpu(mp.global_coord_pool), cpu(mp.sdf_label_pool), cpu(mp.time_pool), cpu(mp.weight_pool))
    lc = O.LoopConfig(sigma=mp.sdf_scale, gradient_decimation=cfg.gradient_decimation,
                      fd_eps=cfg.voxel_size_m * cfg.num_grad_step_ratio, lr=cfg.lr, adam_eps=cfg.adam_eps)
    gen = torch.Generator().manual_seed(1)
    S = pool.global_coord.shape[0]
    bs = 16384
    O.mapping_iters(st, od, pool, [torch.randint(0, S, (bs,), generator=gen)], lc)  # warm-up
    done, t0 = 0, time.perf_counter()
    while done < max_iters and time.perf_counter() - t0 < budget_s:
        O.mapping_iters(st, od, pool, [torch.randint(0, S, (bs,), generator=gen)], lc)
        done += 1
    dt = time.perf_counter() - t0
    return {
        "value": bs * done / dt, "unit": "sampled-points/s", "cores": threads, "kind": "port",
        "sample": f"{done} mapping iterations of the CPU oracle (torch {torch.__version__} eager, fp32) at bs={bs} "
                  f"on the same synthetic map/pool, {dt:.1f} s",
        "ms_per_iter": 1e3 * dt / max(done, 1),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--bs", type=int, default=16384, help="samples per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--freeze-decoder", action="store_true",
                    help="steady-state variant: decoder frozen (freeze_model, slam.py:193-196); not the headline config")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl == RCCL; gloo for single-GPU dry runs)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    local_dev = local_rank % torch.cuda.device_count()  # dry runs may put several ranks on one GPU
    torch.cuda.set_device(local_dev)
    device = f"cuda:{local_dev}"
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(device))
        else:
            dist.init_process_group(args.backend)

    import clid_slam_amd  # noqa: F401
    from clid_slam_amd import HotPathConfig, _lib

    cfg = HotPathConfig()  # == config/run_ncd128.yaml resolved values (SURVEY.md section 8 header)
    cfg.device = device
    cfg.bs = args.bs * world  # global batch; each rank trains on its 16384-sample slice
    nm, dec, mp, scene = build_scene(cfg, device)
    lib = _lib.load()
    M = nm.local_count()

    def sync():
        if dist:
            dist.barrier()
        torch.cuda.synchronize()

    if args.freeze_decoder:
        from clid_slam_amd.tools import freeze_model

        freeze_model(dec)
    mp.mapping(args.warmup)
    sync()
    t0 = time.perf_counter()
    mp.mapping(args.steps)
    sync()
    dt = time.perf_counter() - t0
    if dist:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    losses = mp.last_losses[-1].tolist()

    # ---- roofline leg: per-kernel hipEvent timing on the launch stream (separate pass)
    roof = None
    prof_steps = min(args.steps, 100)
    if rank == 0:
        lib.clid_profile_enable(1)
    mp.mapping(prof_steps)  # every rank takes part (the loop contains collectives when world > 1)
    if rank == 0:
        out = (C.c_double * 5)()
        n = C.c_int(0)
        _lib.check(lib.clid_profile_read(out, C.byref(n), _lib.stream()), "clid_profile_read")
        lib.clid_profile_enable(0)
        ov = out[4]
        hoisted = out[1] > 0.0  # hoisted searches: one search launch per chunk, then decode + Adam per iteration
        names = ("k_train_fused8<2> (decode)" if hoisted else "k_train_fused8", "k_train_fused8<1> (search)",
                 "k_reduce_partials", "k_adam_all")
        ms = [max(out[i] / max(n.value, 1) - ov, 0.0) for i in range(4)]
        if hoisted:  # the search is bracketed once per chunk of <= 32 iterations, reported per iteration
            ms[1] = max(out[1] - ov * math.ceil(n.value / 32), 0.0) / max(n.value, 1)
        decim = cfg.gradient_decimation
        Q = args.bs + 6 * ((args.bs + decim - 1) // decim)
        # algorithmic bytes per launch: search = 688 B per query point (position, 81 bucket
        # probes, the valid probes' positions) + the 24 B pool gather per batch sample; decode fwd+bwd = 316 +
        # 436 B per query point; when the two run as separate kernels the 96 B/query record of winners is
        # written by one and read by the other.  Adam = 256 B per feature row + decoder
        search_b = Q * 688.0 + args.bs * BYTES_POOL_GATHER
        decode_b = Q * (BYTES_FWD_PER_QUERY - 688.0 + BYTES_BWD_PER_QUERY)
        adam_b = BYTES_ADAM_PER_ROW * (M + 1) + 833 * 28.0
        if hoisted:
            alg = [decode_b + Q * 96.0, search_b + Q * 96.0, 0.0, adam_b]
        else:
            alg = [search_b + decode_b, 0.0, 0.0, adam_b]
        dom = max(range(4), key=lambda i: ms[i])
        achieved = alg[dom] / (ms[dom] * 1e-3) / 1e9
        traffic = None  # HBM bytes per launch from the committed PMC passes (same workload only)
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "r01_hbm_traffic.json")))
            if tj["workload"]["bs_per_gpu"] == args.bs and tj["workload"]["decimation"] == decim and names[dom] in tj:
                traffic = tj[names[dom]]["traffic_bytes"]
        except (OSError, KeyError, ValueError):
            pass
        roof = {
            "bound": "hbm", "kernel": names[dom], "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
            "algorithmic_bytes_per_launch": alg[dom], "avg_launch_us": ms[dom] * 1e3,
            "per_kernel_us": {nme: round(m * 1e3, 2) for nme, m in zip(names, ms) if m > 0.0},
            "hoisted_search": hoisted,
            "event_pair_overhead_us": round(ov * 1e3, 2),
            "step_bytes": sum(alg), "step_frac_of_peak": sum(alg) / (dt / args.steps) / 1e9 / HBM_PEAK_GBS,
            "note": "hipEvent-bracketed launches on the launch stream in a separate pass of the same loop, minus the "
                    "measured cost of an empty event pair (rocprofv3 --kernel-trace average for the same kernel: "
                    "profiles/r01_bench_v6_kernel_stats.csv, ~3 us higher because it spans dispatch to completion); "
                    "traffic = offline PMC passes (profiles/r01_hbm_traffic.json); the kernel is bound by dependent-load "
                    "latency and instruction issue (profiles/r01_pmc_v6_summary.txt), not by HBM bandwidth",
        }
    sync()

    base = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        base = cpu_baseline(cfg, nm, dec, mp)

    if rank == 0:
        value = args.bs * world * args.steps / dt
        line = {
            "metric": "sampled-points/sec through SDF-MLP fwd+bwd per mapping iter",
            "value": value, "unit": "sampled-points/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": "run_ncd128.yaml defaults, single-scan mapping loop: fp32, numerical eikonal "
                            f"(decimation {cfg.gradient_decimation}), Adam; synthetic box-room Ouster-128 scan",
                "bs_per_gpu": args.bs, "decoder_frozen": bool(args.freeze_decoder), "global_batch": args.bs * world, "query_points_per_step_per_gpu":
                    args.bs + 6 * ((args.bs + cfg.gradient_decimation - 1) // cfg.gradient_decimation),
                "neural_points_local": M, "pool_samples": int(mp.pool_sample_count), "buffer_size": cfg.buffer_size,
                "parallelism": f"dp{world} (batch sharded, RCCL all-reduce of [decoder|feature] grads)" if world > 1 else "single GPU",
            },
            "final_loss": {"total": losses[0], "bce": losses[1], "eikonal": losses[2]},
            "roofline": roof, "cpu_baseline": base,
        }
        if base:
            line["speedup_vs_cpu_baseline"] = value / base["value"]
        print(json.dumps(line))
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
