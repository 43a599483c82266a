#!/usr/bin/env python3
"""Headline benchmark: sampled points / second through one mapping iteration of CLID-SLAM's SDF training loop
(BASELINE.json `metric`).

    python bench.py [--gpus N --steps K --warmup W] [--config cfg2|cfg3|cfg4]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one iteration of `Mapper.mapping` (utils/mapper.py:642-836): get_batch gathers, query of bs + 6*ceil(bs/10)
points, decode, BCE + numerical eikonal loss, backward, Adam.  Workloads (BASELINE.json `configs`):
  cfg2 (default)  run_ncd128.yaml defaults: fp32, 16384 samples per GPU per step (weak scaling for N > 1)
  cfg3            65536 samples per step, decoder contractions on bf16 MFMA (fp32 accumulation / master weights)
  cfg4            262144 samples per step GLOBAL, sharded over the N ranks (strong scaling), fp32
(cfg5, the SubT_MRS sequence, is a multi-frame run: bench_sequence.py [--gpus N].)
`--gpus N` without a launcher around it (WORLD_SIZE unset) starts the N ranks itself through torch.distributed.run.
The timed region is EXACTLY `--steps` iterations of one `mapping()` call between two barrier+synchronize pairs, all
inputs resident in HBM.  Also reported: the reference's per-frame regime (`mapping(10)` calls, slam.py:187-200, median
of 100), the per-kernel durations / roofline fractions of a profiled pass (hipExtLaunchKernelGGL start/stop events =
the dispatch time stamps rocprofv3 --kernel-trace reads), and the CPU oracle on a bounded sample of the same workload.
Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import ctypes as C
import gc
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# Completion waits by polling instead of interrupts: the timed region is ~0.7 ms and ends in a host wait; an interrupt-driven
# wake-up was measured to arrive 30-60 ms late in about 1 run of 20 (GPU events 0.64 ms, wall clock 58.7 ms:
# `timed_region_split` in the output line).  Must be set before the HSA runtime starts, i.e. before importing torch.
# Single-GPU runs only (or CLID_BENCH_POLL=1): N ranks busy-polling next to RCCL's proxy threads on a host whose core count is not
# known in advance is a first-run risk the multi-GPU line does not need -- its timed region ends in a collective anyway.
def _multi_rank_argv():
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        return True
    for i, a in enumerate(sys.argv):
        if a == "--gpus" and i + 1 < len(sys.argv):
            return sys.argv[i + 1] not in ("0", "1")
        if a.startswith("--gpus="):
            return a.split("=", 1)[1] not in ("0", "1")
    return False


if os.environ.get("CLID_BENCH_POLL", "auto") == "1" or (os.environ.get("CLID_BENCH_POLL", "auto") != "0" and not _multi_rank_argv()):
    os.environ.setdefault("HSA_ENABLE_INTERRUPT", "0")
# the peer-mapped A/B leg of a multi-GPU run: a flag wait gives up after 20 s here (library default 600 s) and the call is
# repeated over RCCL, so a transport that does not work on this node costs seconds, not the run
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")  # (this ROCm's default: kernel arguments in device memory, profiles/r05_kernarg_ab.jsonl)
os.environ.setdefault("CLID_P2P_TIMEOUT_S", "20")

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X spec peak (MI355X_MICROARCH.md); 6290 GB/s is the measured copy ceiling
# Algorithmic bytes per unit (restated in DESIGN.md section 4): per QUERY POINT the forward is
# 1004 B = 688 B of search (position 12, 81 slot probes 324, valid probes' xyz/stamp/index 352) + 316 B of decode
# (6 feature rows + certainties 216, side-effect RMWs 96, sdf 4); the backward is 436 B; a batch sample adds a 24 B
# pool gather; dense Adam moves 256 B per feature row + 28 B per decoder parameter.
B_SEARCH_Q, B_DECODE_FWD_Q, B_BWD_Q = 688.0, 316.0, 436.0
B_POOL_SAMPLE, B_ADAM_ROW, B_ADAM_PARAM = 24.0, 256.0, 28.0

CONFIGS = {
    "cfg2": dict(bs=16384, scaling="weak", decode=1, dtype="f32",
                 what="run_ncd128.yaml defaults, single-scan mapping loop: fp32"),
    "cfg3": dict(bs=65536, scaling="weak", decode=2, dtype="bf16",
                 what="run_ncd128.yaml, 65 536 samples/iter, decoder contractions on bf16 MFMA (fp32 accumulate)"),
    "cfg4": dict(bs=262144, scaling="strong", decode=1, dtype="f32",
                 what="run_ncd128.yaml, 262 144 samples/iter sharded over the ranks, RCCL gradient all-reduce, fp32"),
}


STAMP_FILES = ("clid-slam_amd/csrc", "include/clid_native.h", "clid-slam_amd/mapper.py", "clid-slam_amd/neural_points.py",
               "clid-slam_amd/_lib.py", "clid-slam_amd/decoder.py", "clid-slam_amd/build.py")


def source_stamp():
    """sha256 over the kernel sources, the C header and the host side of mapping(): a committed bench line carries it, and
    tests/test_bench_contract.py fails when the newest profiles/*_bench_cfg2_steps20.json was measured on other sources (works
    on the GPU box, which has no .git)."""
    import hashlib

    h, n = hashlib.sha256(), 0
    for rel in STAMP_FILES:
        path = os.path.join(ROOT, rel)
        files = [os.path.join(path, f) for f in sorted(os.listdir(path))] if os.path.isdir(path) else [path]
        for f in files:
            if os.path.isfile(f):
                h.update(os.path.relpath(f, ROOT).encode())
                h.update(open(f, "rb").read())
                n += 1
    return {"sha16": h.hexdigest()[:16], "files": n}


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


def cpu_baseline(cfg, nm, dec, mp, bs, budget_s=15.0, max_iters=400):
    """The CPU oracle (port of the reference's PyTorch path) on the same map/pool, host cores."""
    from oracle import cpu_ref as O

    threads = min(16, os.cpu_count() or 1)  # slam.py:44 uses 16
    torch.set_num_threads(threads)
    cpu = lambda t: t.detach().cpu().clone()  # noqa: E731
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


def kernel_report(lib, mp, steps, bs_local, decim, M, decode_variant, analytic=False):
    """Profiled pass: average dispatch duration per kernel + roofline fraction on the section-8(d) bytes."""
    from clid_slam_amd import _lib

    prof = lib.clid_profile_create()  # handed to the loop through clid_train_args.prof (no state inside the library)
    mp._prof = prof
    out = (C.c_double * 6)()
    n = C.c_int(0)
    try:
        mp.mapping(steps)
        _lib.check(lib.clid_profile_read(prof, out, C.byref(n), _lib.stream()), "clid_profile_read")
    finally:
        mp._prof = None
        lib.clid_profile_destroy(prof)
    iters = max(n.value, 1)
    Q = bs_local + 6 * ((bs_local + decim - 1) // decim)
    hoisted = out[1] > 0.0
    n_tasks_est = int(lib.clid_train_search_tasks(bs_local, 0, decim, 1))  # <= 2048 tiles: the launch that also numbers the tiles
    dname = {0: "k_train_fused8<2> (decode, 16 lanes/query)", 1: "k_decode_tile<fp32 MFMA>", 2: "k_decode_tile<bf16 MFMA>"}[decode_variant]
    if analytic:  # loss.numerical_grad_on: False -- one launch per iteration: search + decode + analytic d sdf / d x + backward through it
        rows = [("k_train_analytic (decode + analytic eikonal + backward through it, 16 lanes/query%s)" % ("" if hoisted else "; search inside"),
                 out[0] / iters, bs_local * (1860.0 - (B_SEARCH_Q if hoisted else 0.0)),
                 "1860 B/sample (SURVEY 8d: 1440 per query fwd+bwd + 24 pool gather + 12 gradient write + 384 second-order feature-gradient pass) x %d samples" % bs_local),
                ("k_search_tiles / k_search_tasks (cell-directory search of the plain tasks, per iteration of a <=32-iteration launch)",
                 out[1] / iters, bs_local * (B_SEARCH_Q + B_POOL_SAMPLE), "688 B/query + 24 B/sample x %d samples" % bs_local),
                ("k_adam_all", out[3] / iters, B_ADAM_ROW * (M + 1) + 833 * B_ADAM_PARAM, "256 B/row x %d rows + 28 B x 833" % (M + 1))]
    else:
      rows = [
        (dname if hoisted else "k_train_fused8<0> (search+decode)", out[0] / iters,
         Q * (B_DECODE_FWD_Q + B_BWD_Q) if hoisted else Q * (B_SEARCH_Q + B_DECODE_FWD_Q + B_BWD_Q) + bs_local * B_POOL_SAMPLE,
         "752 B/query (316 fwd after the search + 436 bwd) x %d query points" % Q if hoisted else "1440 B/query + 24 B/sample"),
        (("k_search_tiles (cell-directory search + tile numbering" if (n_tasks_est + 1) // 2 <= 2048 else
          "k_search_tasks (cell-directory search") + ", per iteration of a <=32-iteration launch)", out[1] / iters,
         Q * B_SEARCH_Q + bs_local * B_POOL_SAMPLE, "688 B/query x %d + 24 B x %d samples" % (Q, bs_local)),
        ("k_reduce_partials (+ pack of the touched rows)", out[2] / iters, 0.0, ""),
        ("k_touch_bits + k_touch_scan (per iteration of a chunk)", out[5] / iters, 0.0, ""),
        ("k_adam_all", out[3] / iters, B_ADAM_ROW * (M + 1) + 833 * B_ADAM_PARAM, "256 B/row x %d rows + 28 B x 833" % (M + 1)),
    ]
    kernels = []
    for name, ms, alg, how in rows:
        if ms <= 0.0:
            continue
        gbs = alg / (ms * 1e-3) / 1e9 if alg else None
        kernels.append({"kernel": name, "avg_us": round(ms * 1e3, 3), "algorithmic_bytes": alg, "bytes_rule": how,
                        "achieved_GBs": gbs and round(gbs, 1), "frac_of_hbm_peak": gbs and round(gbs / HBM_PEAK_GBS, 4)})
    return kernels, iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="cfg2", choices=sorted(CONFIGS))
    ap.add_argument("--bs", type=int, default=None, help="override the samples per step (per GPU for weak scaling)")
    ap.add_argument("--scaling", default=None, choices=("weak", "strong"))
    ap.add_argument("--decode", type=int, default=None, choices=(0, 1, 2), help="override the decode kernel (clid_train_args.decode_variant)")
    ap.add_argument("--frame-calls", type=int, default=100, help="repetitions of mapping(10) for the per-frame regime (0 = skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--analytic", action="store_true",
                    help="loss.numerical_grad_on: False (utils/mapper.py:695-696): analytic d sdf / d x on EVERY sample, backward through it")
    ap.add_argument("--layer-norm", action="store_true", help="layer_norm_on: True (run_SubT_MRS.yaml:27) instead of the ncd128 default")
    ap.add_argument("--freeze-decoder", action="store_true",
                    help="steady-state variant: decoder frozen (freeze_model, slam.py:193-196); not the headline config")
    ap.add_argument("--diag", action="store_true", help="extra host time stamps around the timed region (start latency of the "
                    "first launch, poll vs synchronize); adds two event polls to the region: measurement aid only")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl == RCCL; gloo for single-GPU dry runs)")
    ap.add_argument("--exchange-ab", default=os.environ.get("CLID_BENCH_EXCHANGE_AB", "rccl"), choices=("none", "rccl", "all"),
                    help="multi-GPU: extra timed legs in the same processes -- none; rccl = dense and compact payload over RCCL "
                         "(default); all = also the peer-mapped transport (csrc/p2p.hip: never run across devices yet, opt-in)")
    args = ap.parse_args()
    wl = dict(CONFIGS[args.config])
    if args.bs:
        wl["bs"] = args.bs
    if args.scaling:
        wl["scaling"] = args.scaling
    if args.decode is not None:
        wl["decode"] = args.decode
    elif "CLID_DECODE" in os.environ and args.config == "cfg2":
        wl["decode"] = int(os.environ["CLID_DECODE"])

    if args.gpus < 1:
        raise SystemExit(f"--gpus {args.gpus}")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks (one process per GPU) ourselves
        from clid_slam_amd.dist import respawn_under_torchrun

        raise SystemExit(respawn_under_torchrun(os.path.abspath(__file__), sys.argv[1:], args.gpus, args.backend))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the line would report the wrong n_gpus")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    if world > 1 and args.backend == "nccl" and torch.cuda.device_count() < world:
        raise SystemExit(f"--gpus {world} over RCCL but only {torch.cuda.device_count()} GPU(s) visible (one device per rank)")
    local_dev = local_rank % torch.cuda.device_count()  # gloo dry runs may put several ranks on one GPU
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
    cfg.layer_norm_on = bool(args.layer_norm)
    if args.analytic:
        cfg.numerical_grad, cfg.gradient_decimation = False, 1
    cfg.device = device
    if wl["scaling"] == "strong":
        if wl["bs"] % world:
            raise SystemExit(f"global batch {wl['bs']} is not divisible by {world} ranks")
        bs_global, bs_local = wl["bs"], wl["bs"] // world
    else:
        bs_global, bs_local = wl["bs"] * world, wl["bs"]
    cfg.bs = bs_global
    nm, dec, mp, scene = build_scene(cfg, device)
    lib = _lib.load()
    rccl_ranks = 0
    if dist:  # RCCL communicator behind the C ABI (csrc/comm.hip): the sharded loop then needs no Python per iteration
        comm = _lib.rccl_comm(dist)
        rccl_ranks = int(lib.clid_comm_size(comm)) if comm is not None else 0
        if args.backend == "nccl" and rccl_ranks != world:
            raise SystemExit(f"rank {rank}: the RCCL communicator behind the C ABI has {rccl_ranks} ranks, expected {world}")
    M = nm.local_count()
    decim = cfg.gradient_decimation

    def sync():
        if dist:
            dist.barrier()
        torch.cuda.synchronize()

    if args.freeze_decoder:
        from clid_slam_amd.tools import freeze_model

        freeze_model(dec)
    mp.decode_variant = wl["decode"]
    mp.reserve(max(args.steps, args.warmup, 10))  # workspaces sized once: nothing is allocated inside a timed region
    # No collector pause inside the 0.7 ms region (what `timeit` does too): a full collection of this process' ~10^6 objects
    # takes several ms -- one run in ~25 showed 5 ms of wall clock around 0.65 ms of GPU time.  Collected and switched off
    # BEFORE the warm-up: a collection between warm-up and timed region idles the GPU for ~100 ms and its clocks drop again
    # (measured: 0.044 instead of 0.035 ms per step).
    gc.collect()
    gc.disable()

    def timed_region():
        """W untimed warm-up steps, then EXACTLY K steps of one mapping() call between two barrier + synchronize pairs;
        returns (seconds = max over the ranks, split of the region on this rank)."""
        for _ in range(args.warmup):  # W untimed steps, one per call: the GPU's clocks ramp over several hundred us of work
            mp.mapping(1)             # (a single 5-iteration call leaves the first timed call 14 % slow: tools/warm_clocks.py)
            torch.cuda.synchronize()  # each step complete before the next is enqueued: five calls queued back to back leave
                                      # the first timed call 15-20 us (3 %) slow (tools/warm_clocks.py five / fivesync)
        sync()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()  # (a diagnostic time stamp on the idle stream, not part of the K steps: enqueued before the clock starts)
        t0 = time.perf_counter()
        t_start = None
        if args.diag:
            while not ev0.query():
                pass
            t_start = time.perf_counter() - t0
        mp.mapping(args.steps)
        ev1.record()
        t_enq = time.perf_counter() - t0
        # The region ends in the contract's barrier + torch.cuda.synchronize().  With polling completion waits
        # (HSA_ENABLE_INTERRUPT=0, single-GPU runs: top of this file) the synchronize IS a poll and returns within a few us of the
        # last kernel; an event-query loop in front of it -- rounds 3-5 had one against interrupt-driven waits that returned
        # 30-60 ms late about 1 run in 20 -- only adds its own 22 us to the region (tools/region_end_timing.py: 0.515 vs 0.493 ms
        # per mapping(20) call, alternating on one box).  It stays for interrupt-driven waits (multi-GPU runs).
        t_poll = None
        if os.environ.get("HSA_ENABLE_INTERRUPT") != "0":
            while not ev1.query():
                pass
            t_poll = time.perf_counter() - t0
        sync()
        dt = time.perf_counter() - t0
        split = {"host_enqueue_ms": 1e3 * t_enq, "gpu_ms": float(ev0.elapsed_time(ev1)), "wall_ms": 1e3 * dt,
                 "poll_done_ms": None if t_poll is None else 1e3 * t_poll}
        if t_start is not None:
            split["first_event_done_ms"] = 1e3 * t_start
        if dist:
            t = torch.tensor([dt], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, split

    def exchange_report(dt):
        ex = mp.last_exchange or {}
        return {"ms_per_step": 1e3 * dt / args.steps, "value": bs_global * args.steps / dt,
                "mode": ex.get("mode"), "transport": ex.get("transport"), "bytes_per_iter": ex.get("bytes_per_iter"),
                "dense_bytes_per_iter": ex.get("dense_bytes_per_iter"), "p2p_fallbacks": int(mp.p2p_fallbacks)}

    # ---- the headline leg: the default exchange -- RCCL all-reduce (north_star), dense or compact by map size
    dt, timed_split = timed_region()
    losses = mp.last_losses[-1].tolist()
    legs = None
    if dist:
        # ---- A/B of the gradient exchange in the SAME processes (one SCALE run answers which transport to keep): every
        # leg is a full W-warm-up + K-step region like the headline.  A timed-out flag wait of the peer-mapped transport
        # does not abort: Mapper.mapping restores, falls back to RCCL and the leg reports `p2p_fallbacks` (and the
        # transport it actually ended on).
        headline = exchange_report(dt)
        legs = {}
        plan = (("rccl_dense", "dense", "rccl"), ("rccl_compact", "compact", "rccl"), ("p2p_compact", "compact", "p2p"))
        plan = tuple(p for p in plan if args.exchange_ab == "all" or (args.exchange_ab == "rccl" and p[2] == "rccl"))
        for name, mode, transport in plan:
            if headline["mode"] == mode and transport == "rccl":
                legs[name] = dict(headline, headline=True)
                continue
            mp.exchange_mode, mp.exchange_transport = mode, transport
            leg_dt, _ = timed_region()
            legs[name] = exchange_report(leg_dt)
        mp.exchange_mode, mp.exchange_transport = None, None
        if any(v["p2p_fallbacks"] for v in legs.values()):
            legs["p2p_error"] = getattr(mp, "last_p2p_error", None)
        mp.mapping(1)  # (the remaining legs of this file run on the default exchange again)
    gc.enable()

    # ---- the reference's per-frame regime: mapping(10) per scan (slam.py:187-200), each call timed on its own
    frame = None
    if args.frame_calls > 0:
        per_call = []
        for _ in range(args.frame_calls):
            sync()
            t1 = time.perf_counter()
            mp.mapping(10)
            torch.cuda.synchronize()
            per_call.append(time.perf_counter() - t1)
        med = statistics.median(per_call)
        if dist:
            t = torch.tensor([med], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            med = float(t.item())
        frame = {"iters_per_call": 10, "calls": args.frame_calls, "median_ms_per_call": 1e3 * med,
                 "ms_per_step": 1e2 * med, "value": bs_global * 10 / med, "p10_ms_per_call": 1e3 * sorted(per_call)[len(per_call) // 10],
                 "note": "one Mapper.mapping(10) call per repetition incl. index draw, workspace reset, hoisted search of the "
                         "10 batches, 10 x (decode, Adam), write-back to the global map; host-synchronised around each call"}

    # ---- roofline leg: dispatch time stamps of every kernel in a separate pass of the same loop
    kernels, roof = None, None
    prof_steps = min(args.steps, 100)
    if rank == 0:
        kernels, n_prof = kernel_report(lib, mp, prof_steps, bs_local, decim, M, wl["decode"], analytic=args.analytic)
    else:
        mp.mapping(prof_steps)  # every rank takes part (the loop contains collectives when world > 1)
    if rank == 0:
        dom = max(kernels, key=lambda k: k["avg_us"])
        traffic, second, stamp = None, None, None  # from the committed PMC passes (same workload and kernel only)
        for fn in ("r06_hbm_traffic.json", "r05_hbm_traffic.json", "r04_hbm_traffic.json"):
            try:
                tj = json.load(open(os.path.join(ROOT, "profiles", fn)))
            except (OSError, ValueError):
                continue
            stamp = dict(tj.get("source", {}), file="profiles/" + fn)
            wk = tj.get("workload", {})
            ent = tj.get(dom["kernel"]) if (wk.get("bs_per_gpu") == bs_local and wk.get("decimation") == decim) else None
            if ent is None and bs_local == 65536:
                ent = tj.get("cfg3", {}).get(dom["kernel"])
            if ent:
                traffic = ent.get("traffic_bytes")
                iss, res = ent.get("issue"), ent.get("resources")
                if iss and iss.get("valu_port_cycles"):
                    # the roof that binds at these sizes is not HBM: VALU / matrix-port occupancy of the launch and the waves a
                    # SIMD can keep resident (MI355X_MICROARCH.md: wave64 VALU = 2 cycles on a SIMD-32, fp32 16x16x4 MFMA 32
                    # cycles per SIMD; 256 CUs x 4 SIMDs, 2.4 GHz)
                    simds, clock = 1024, 2.4e9
                    frac = iss["valu_port_cycles"] / (simds * clock * dom["avg_us"] * 1e-6)
                    wait = (iss["wait_any"] / iss["wave_cycles"]) if iss.get("wave_cycles") else None
                    second = {"issue_frac": round(frac, 4), "valu_port_cycles_per_launch": iss["valu_port_cycles"],
                              "valu_insts": iss["valu_insts"], "mfma_busy_cycles": iss["mfma_busy_cycles"], "salu_insts": iss.get("salu_insts"),
                              "lds_insts": iss.get("lds_insts"), "vmem_insts": iss.get("vmem_insts"), "waves": iss.get("waves"),
                              "wait_any_over_wave_cycles": wait and round(wait, 3),
                              "occupancy_waves_per_simd": res and res.get("occupancy_waves_per_simd"), "resources": res,
                              "rule": "issue_frac = (2 x (VALU - MFMA instructions) + MFMA busy cycles) / (1024 SIMDs x 2.4 GHz x the "
                                      "launch duration measured in THIS run); instruction counts from the offline PMC pass"}
            break
        step_bytes = sum(k["algorithmic_bytes"] for k in kernels)
        roof = {
            # `bound` names the roof `achieved / peak` is PRICED against -- the contract offers "hbm" or "mfma", and of the two the
            # path is the HBM-side one (byte gathers, 10 TFLOP/s of matrix work) -- not a claim that HBM bandwidth limits the launch:
            # counter traffic is BELOW the algorithmic bytes (the map is cache-resident).  What limits it is in `limiter` / `issue`.
            "bound": "hbm", "bound_is_the_pricing_roof_not_the_limiter": True,
            "kernel": dom["kernel"], "achieved": dom["achieved_GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": dom["frac_of_hbm_peak"], "traffic": traffic,
            "limiter": (None if second is None else
                      ("latency: %s waves per SIMD resident (register / LDS limited), VALU + matrix port busy %.0f %% of the launch, "
                       "waves parked %.0f %% of their cycles" % (second["occupancy_waves_per_simd"], 100 * second["issue_frac"],
                                                                 100 * (second["wait_any_over_wave_cycles"] or 0.0)))
                      if second["issue_frac"] < 0.6 else "VALU / matrix issue"),
            "issue": second, "counters_from": stamp,
            "algorithmic_bytes_per_launch": dom["algorithmic_bytes"], "bytes_rule": dom["bytes_rule"],
            "avg_launch_us": dom["avg_us"], "kernels": kernels, "profiled_steps": n_prof,
            "step_bytes": step_bytes, "step_frac_of_peak": step_bytes / (dt / args.steps) / 1e9 / HBM_PEAK_GBS,
            "note": "avg_launch_us = mean dispatch begin->end of the kernel (hipExtLaunchKernelGGL start/stop events on the "
                    "launch stream, the clock rocprofv3 --kernel-trace reports; committed trace: profiles/r06_*_kernel_stats.csv); "
                    "achieved = SURVEY section 8(d) algorithmic bytes of the launch / that duration; traffic = offline PMC passes "
                    "(the file `counters_from` names, FETCH_SIZE / WRITE_SIZE as the guide corrects them); the loop is bound by dependent-launch latency "
                    "and the memory-side atomic rate at this batch size, not by HBM bandwidth (DESIGN.md section 6)",
        }
    sync()

    base = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        base = cpu_baseline(cfg, nm, dec, mp, min(bs_local, 16384))

    if rank == 0:
        value = bs_global * args.steps / dt
        line = {
            "metric": "sampled-points/sec through SDF-MLP fwd+bwd per mapping iter",
            "value": value, "unit": "sampled-points/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": wl["scaling"], "vs_baseline": None,
            "dtype": wl["dtype"], "data": "synthetic",
            "config": {
                "workload": f"{args.config}: {wl['what']}; {'analytic eikonal on every sample' if args.analytic else f'numerical eikonal (decimation {decim})'}, Adam; synthetic box-room "
                            "Ouster-128 scan",
                "bs_per_gpu": bs_local, "global_batch": bs_global, "decoder_frozen": bool(args.freeze_decoder), "layer_norm_on": bool(args.layer_norm),
                "decode_kernel": wl["decode"], "query_points_per_step_per_gpu": bs_local + 6 * ((bs_local + decim - 1) // decim),
                "neural_points_local": M, "pool_samples": int(mp.pool_sample_count), "buffer_size": cfg.buffer_size,
                "parallelism": f"dp{world} (batch sharded, RCCL all-reduce of [decoder|feature] grads)" if world > 1 else "single GPU",
                "rccl_ranks_in_c_abi": rccl_ranks,
                "host_cores": os.cpu_count(), "completion_wait": "poll" if os.environ.get("HSA_ENABLE_INTERRUPT") == "0" else "interrupt",
                "gradient_exchange": legs,
            },
            "final_loss": {"total": losses[0], "bce": losses[1], "eikonal": losses[2]},
            "per_frame_regime": frame, "roofline": roof, "cpu_baseline": base,
            "timed_region_split": timed_split,  # host time to enqueue the K steps | GPU time between events | wall clock
            "source_stamp": source_stamp(),
        }
        if base:
            line["speedup_vs_cpu_baseline"] = value / base["value"]
        print(json.dumps(line))
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
