"""`Mapper` with the reference's interface (utils/mapper.py:35): the online SDF training loop.

`mapping(iter_count)` (utils/mapper.py:620-862) is the hot loop.  Here each iteration is the fused
HIP sequence of csrc/train.hip (forward over batch + finite-difference points, loss, backward,
Adam) driven through the C ABI; nothing in it runs in PyTorch autograd, and it raises if the HIP
library or the GPU is missing.  `get_batch`, `sdf` and `get_numerical_gradient` keep the
reference's signatures for callers that use the un-fused sequence.

Multi-GPU (new; the reference is single-GPU): with `torch.distributed` initialised, `config.bs` is
the GLOBAL batch.  Every rank draws the same index sequence (same seed), trains on its contiguous
slice, and the fused gradient buffer [decoder | features] is all-reduced (RCCL, SUM) before the
identical Adam step on every replica; certainty increments (SUM) and update stamps (MAX) are merged
once per `mapping()` call.
"""
from __future__ import annotations

import ctypes as C
import os
import sys

import numpy as np
import torch

from . import _lib

POOL_GATE_DEFAULT = "0"  # process_frame: hold the pool compaction back until the map growth's voxel pass is through (A/B)


def _dist():
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized() and (
            dist.get_world_size() > 1 or os.environ.get("CLID_DIST_SINGLE") == "1"):  # (debug aid: sharded path with one rank)
        return dist
    return None


class Mapper:
    def __init__(self, config, dataset, neural_points, local_point_cloud_map, geo_mlp, sem_mlp=None, color_mlp=None):
        self.config = config
        self.silence = config.silence
        self.dataset = dataset
        self.neural_points = neural_points
        self.local_point_cloud_map = local_point_cloud_map
        self.geo_mlp = geo_mlp
        self.sem_mlp = sem_mlp
        self.color_mlp = color_mlp
        self.device = config.device
        self.dtype = config.dtype
        self.used_poses = None
        # utils/mapper.py:57-69: analytic gradient only when the numerical one is off
        self.require_gradient = bool(
            config.ekional_loss_on or getattr(config, "proj_correction_on", False) or getattr(config, "consistency_loss_on", False)
        )
        if config.numerical_grad and not getattr(config, "proj_correction_on", False) and not getattr(config, "consistency_loss_on", False):
            self.require_gradient = False
        self.total_iter: int = 0
        self.sdf_scale = config.logistic_gaussian_ratio * config.sigma_sigmoid_m
        from .data_sampler import DataSampler

        self.sampler = DataSampler(config)  # utils/mapper.py:74
        self.ray_sample_count = 1 + config.surface_sample_n + config.free_behind_n + config.free_front_n
        self.new_idx = None
        self.ba_done_flag = False
        self.adaptive_iter_offset = 0
        dev, dt = self.device, self.dtype
        # data pool (utils/mapper.py:84-97)
        self.coord_pool = torch.empty((0, 3), device=dev, dtype=dt)
        self.global_coord_pool = torch.empty((0, 3), device=dev, dtype=dt)
        self.sdf_label_pool = torch.empty((0), device=dev, dtype=dt)
        self.color_pool = None
        self.sem_label_pool = None
        self.normal_label_pool = None
        self.weight_pool = torch.empty((0), device=dev, dtype=dt)
        self.time_pool = torch.empty((0), device=dev, dtype=torch.int)
        self.pool_sample_count = 0
        self.cur_sample_count = 0
        # fused-loop state
        self.last_losses = None  # [iters,4] device tensor: total, bce, eikonal, -
        self._ws = None
        self._gen = None
        self._seed = int(getattr(config, "seed", 42))
        # per-call switches of the fused loop (None = the process defaults _lib.DECODE_VARIANT / _lib.PIPELINE, i.e. the
        # environment variables CLID_DECODE / CLID_PIPELINE): the decode kernel (0 VALU, 1 tile fp32 MFMA, 2 tile bf16
        # MFMA) and the schedule (1 hoisted searches, 0 one fused launch per iteration).  They travel in clid_train_args.
        self.decode_variant = None
        self.pipeline = None
        self.last_exchange = None  # world > 1: what the last mapping() call all-reduced (floats, mode)
        # world > 1: how the per-iteration gradient payload travels.  transport None = RCCL unless CLID_P2P=1 ("rccl" / "p2p":
        # peer-mapped buffers, csrc/p2p.hip); mode None = by map size, CLID_SPARSE ("dense" / "compact")
        self.exchange_transport = None
        self.exchange_mode = None
        self.p2p_fallbacks = 0     # calls repeated over RCCL after a timed-out flag wait of the peer-mapped exchange

    def reserve(self, iter_count: int):
        """Size the cached workspaces of `mapping()` for calls of up to `iter_count` iterations on the current local
        map and batch size (optional: `mapping()` grows them on demand; a caller that wants allocation-free calls --
        e.g. a timed region -- reserves once)."""
        lib = _lib.load()
        cfg, nm = self.config, self.neural_points
        dev = nm.local_geo_features.device
        dist = _dist()
        bs_local = int(cfg.bs) // (dist.get_world_size() if dist else 1)
        self._loop_buffers(nm.local_geo_features.shape[0], iter_count, dev)
        need = int(lib.clid_train_workspace_floats(bs_local, max(int(cfg.gradient_decimation), 1), 1))
        if getattr(self, "_ws", None) is None or self._ws.numel() < need or self._ws.device != dev:
            self._ws = torch.empty(need, device=dev, dtype=torch.float32)
        n_idx = iter_count * int(cfg.bs)
        if getattr(self, "_idx_buf", None) is None or self._idx_buf.numel() < n_idx or self._idx_buf.device != dev:
            self._idx_buf = torch.empty(n_idx, device=dev, dtype=torch.int64)
        need = int(lib.clid_mapping_prep_workspace_bytes(iter_count, int(cfg.bs)))  # batch ordering: unsorted draws | keys
        if getattr(self, "_sort_ws", None) is None or self._sort_ws.numel() < need or self._sort_ws.device != dev:
            self._sort_ws = torch.empty(need + 256, device=dev, dtype=torch.uint8)
        if self.pool_sample_count > 0 and self.global_coord_pool.is_cuda and self.global_coord_pool.dtype == torch.float32:
            # first launch of the draw / ordering kernels into the reserved buffers (code-object load, LDS configuration)
            _lib.check(lib.clid_mapping_prep(None, 0, self._idx_buf.data_ptr(), iter_count, int(cfg.bs), 0,
                                             int(self.pool_sample_count), None, 0, 0, 0, self.global_coord_pool.data_ptr(),
                                             float(nm.resolution), self._sort_ws.data_ptr(), 0, 0, self._eik_decimation(), _lib.stream()),
                       "clid_mapping_prep")

    DEC_COPIES = 32  # sharded dense exchange: copies the decode blocks add their decoder-gradient sums to (include/clid_native.h)

    def _loop_buffers(self, n_rows: int, iters: int, dev, zero: bool = True):
        """Views [grad | m | v | m_mlp | v_mlp | losses] of one flat fp32 buffer, zeroed (zero=False: the caller resets
        `self._flat[:self._flat_used]` itself, see `_prepare_call`).  The buffer is cached and only re-allocated when it
        has to grow, so a steady-state `mapping()` call allocates nothing; `last_losses` therefore stays valid until the
        next call."""
        n_feat = n_rows * _lib.F
        # (several ranks: DEC_COPIES x 848 floats behind the accumulation rows -- the decoder-gradient copies of the dense exchange,
        # clid_train_args.dec_copies: inside `grad`, so the all-reduce of the buffer carries them)
        tail = self.DEC_COPIES * 848 if _dist() is not None else 0
        sizes = (_lib.GRAD_FEAT_OFFSET16 + n_rows * _lib.GRAD_ROW16 + tail, n_feat, n_feat, 848, 848, iters * 4)
        total = sum(sizes)
        flat = getattr(self, "_flat", None)
        if flat is None or flat.numel() < total or flat.device != torch.device(dev):
            flat = torch.empty(int(total * 1.25) + 4096, device=dev, dtype=torch.float32)
            self._flat = flat
        self._flat_used = (total + 3) & ~3
        if zero:
            flat[:total].zero_()
        key = (n_rows, iters, flat.data_ptr(), tail)
        cache = self.__dict__.setdefault("_flat_views", {})  # the six views of a layout (slicing costs ~4 us each); a few
        hit = cache.get(key)                                   # layouts alternate (frame 0 / steady state / warm-up)
        if hit is not None:
            return list(hit)
        if len(cache) >= 8:
            cache.clear()
        out, off = [], 0
        for n in sizes:
            out.append(flat[off:off + n])
            off += n
        out[3], out[4] = out[3][:_lib.MLP_PARAMS], out[4][:_lib.MLP_PARAMS]
        out[5] = out[5].view(iters, 4)
        cache[key] = tuple(out)
        return out

    # ------------------------------------------------------------------ a1
    def _draw_index(self, iters: int, bs: int) -> torch.Tensor:
        """[iters, bs] int64 batch indices composed as utils/mapper.py:473-500 (device RNG)."""
        dev = self.global_coord_pool.device
        gen = getattr(self, "_gen", None)
        if gen is None or gen.device != torch.device(dev):
            self._gen = torch.Generator(device=dev)
            self._gen.manual_seed(int(getattr(self, "_seed", getattr(self.config, "seed", 42))))
        use_new = (
            self.config.bs_new_sample > 0 and self.new_idx is not None and self.new_idx.shape[0] > 0
            and not getattr(self.dataset, "lose_track", False) and not getattr(self.dataset, "stop_status", False)
        )
        if use_new:
            bs_new = min(self.new_idx.shape[0], self.config.bs_new_sample)
            hist = torch.randint(0, self.pool_sample_count, (iters, bs - bs_new), device=dev, generator=self._gen)
            pick = torch.randint(0, self.new_idx.shape[0], (iters, bs_new), device=dev, generator=self._gen)
            return torch.cat((hist, self.new_idx[pick]), dim=1).contiguous()
        buf = getattr(self, "_idx_buf", None)  # cached by reserve(): no allocation in steady state
        if buf is not None and buf.numel() >= iters * bs and buf.device == torch.device(dev):
            out = buf[: iters * bs].view(iters, bs)
            return torch.randint(0, self.pool_sample_count, (iters, bs), device=dev, generator=self._gen, out=out)
        return torch.randint(0, self.pool_sample_count, (iters, bs), device=dev, generator=self._gen)

    SORT_BATCH_MIN_ITERS = 6

    def _eik_decimation(self) -> int:
        """The stride of the loop's eikonal subset coord[::d] (utils/mapper.py:700-704): the batch ordering keeps the draws of
        those positions on those positions (clid_mapping_prep `decimation`); 1 when every sample / none is in the subset."""
        cfg = self.config
        on = (cfg.ekional_loss_on and cfg.weight_e > 0 and cfg.numerical_grad and not getattr(cfg, "proj_correction_on", False)
              and not getattr(cfg, "consistency_loss_on", False) and os.environ.get("CLID_ORDER_CLASSES", "1") != "0")
        return max(int(cfg.gradient_decimation), 1) if on else 1  # (CLID_ORDER_CLASSES=0: one class, the plain order -- A/B)

    def _prepare_call(self, iters: int, bs: int, n_rows: int, dev, lib, col0: int = 0, ncols: int = 0):
        """Workspace reset + batch draw of one `mapping()` call in ONE launch (`clid_mapping_prep`): the composition rule
        of `_draw_index` (utils/mapper.py:473-500) with a counter-based generator keyed on (seed, call number, position),
        so every rank of a multi-GPU run draws the same batches -- and draws / orders only the columns [col0, col0 + ncols)
        it trains on (its shard; widened to whole ordering segments).  Returns (loop buffers, index_seq [iters, bs])."""
        bufs = self._loop_buffers(n_rows, iters, dev, zero=False)
        buf = getattr(self, "_idx_buf", None)
        if buf is None or buf.numel() < iters * bs or buf.device != torch.device(dev):
            buf = self._idx_buf = torch.empty(iters * bs, device=dev, dtype=torch.int64)
        # (everything in front of the call's first launch is device idle time: views and sizes are cached)
        iv = self.__dict__.get("_idx_view")
        if iv is None or iv[0] != (buf.data_ptr(), iters, bs):
            iv = self._idx_view = ((buf.data_ptr(), iters, bs), buf[: iters * bs].view(iters, bs))
        index_seq = iv[1]
        use_new = (
            self.config.bs_new_sample > 0 and self.new_idx is not None and self.new_idx.shape[0] > 0
            and not getattr(self.dataset, "lose_track", False) and not getattr(self.dataset, "stop_status", False)
        )
        bs_new, new_ptr, n_new = 0, None, 0
        if use_new:
            new_idx = _lib.require_cuda(self.new_idx, "new_idx", torch.int64)
            bs_new, new_ptr, n_new = min(new_idx.shape[0], self.config.bs_new_sample, bs), new_idx.data_ptr(), new_idx.shape[0]
        self._draw_calls = getattr(self, "_draw_calls", 0) + 1
        seed = int(getattr(self, "_seed", getattr(self.config, "seed", 42))) & 0xFFFFFFFFFFFFFFFF
        # spatially ordered batches (the same draws, each 16 384-sample segment written in Morton order of the samples'
        # voxels by a second launch; both run while the host assembles the loop's arguments).  CLID_SORT_BATCH=0 keeps the
        # draw order.
        coord_ptr, sort_ptr = None, None
        sort_mode = os.environ.get("CLID_SORT_BATCH", "auto")  # auto: calls long enough for the kernels' gain (3 us per
        # iteration at 16 384 samples) to exceed the exposed part of the ordering launch (28 us, mostly under the host's
        # argument assembly); 1: always; 0: never
        if ((sort_mode == "1" or (sort_mode != "0" and iters >= self.SORT_BATCH_MIN_ITERS))
                and self.global_coord_pool.dtype == torch.float32):
            sizes = self.__dict__.setdefault("_sort_need", {})
            need = sizes.get((iters, bs))
            if need is None:
                if len(sizes) > 64:
                    sizes.clear()
                need = sizes[(iters, bs)] = int(lib.clid_mapping_prep_workspace_bytes(iters, bs))
            ws = getattr(self, "_sort_ws", None)
            if ws is None or ws.numel() < need or ws.device != torch.device(dev):
                ws = self._sort_ws = torch.empty(int(need * 1.25) + 256, device=dev, dtype=torch.uint8)
            coord_ptr = _lib.require_cuda(self.global_coord_pool, "global_coord_pool", torch.float32).data_ptr()
            sort_ptr = ws.data_ptr()
        _lib.check(lib.clid_mapping_prep(self._flat.data_ptr(), self._flat_used, buf.data_ptr(), iters, bs, bs_new,
                                         int(self.pool_sample_count), new_ptr, n_new, seed, self._draw_calls, coord_ptr,
                                         float(self.neural_points.resolution), sort_ptr, int(col0), int(ncols),
                                         self._eik_decimation(), _lib.stream()),
                   "clid_mapping_prep")
        return bufs, index_seq

    def get_batch(self, global_coord=False):
        """utils/mapper.py:473-523: 7-tuple (coord, sdf_label, ts, normal, sem, color, weight)."""
        index = self._draw_index(1, self.config.bs)[0]
        coord = self.global_coord_pool[index, :] if global_coord else self.coord_pool[index, :]
        sem = self.sem_label_pool[index] if self.sem_label_pool is not None else None
        col = self.color_pool[index] if self.color_pool is not None else None
        nrm = self.normal_label_pool[index, :] if self.normal_label_pool is not None else None
        return coord, self.sdf_label_pool[index], self.time_pool[index], nrm, sem, col, self.weight_pool[index]

    # config.main_loss_type -> clid_train_args.main_loss_type (utils/mapper.py:751-767)
    MAIN_LOSS_TYPES = {"bce": 0, "sdf_l1": 1, "sdf_l2": 2, "zhong": 3}

    # ------------------------------------------------------------------ a11
    def _check_fused_config(self):
        c = self.config
        bad = []
        if getattr(c, "semantic_on", False):
            bad.append("semantic_on")
        if getattr(c, "color_on", False):
            bad.append("color_on")
        if getattr(c, "consistency_loss_on", False) and (not c.weighted_first or c.main_loss_type != "bce" or self.ba_done_flag
                                                          or getattr(c, "ekional_add_to", "all") != "all" or _dist() is not None):
            bad.append("consistency_loss_on with weighted_first: False / a loss other than bce / ekional_add_to != all / ba_done_flag "
                       "/ several ranks")
        if getattr(c, "proj_correction_on", False) and (not c.weighted_first or c.main_loss_type != "bce"
                                                         or getattr(c, "ekional_add_to", "all") != "all"):
            bad.append("proj_correction_on with weighted_first: False / a loss other than bce / ekional_add_to != all")
        if c.main_loss_type not in self.MAIN_LOSS_TYPES:
            sys.exit("Please choose a valid loss type")  # utils/mapper.py:766-767
        add_to = getattr(c, "ekional_add_to", "all")
        if c.ekional_loss_on and add_to not in ("all", "surface", "freespace"):
            bad.append(f"ekional_add_to={add_to}")
        if not getattr(c, "opt_adam", True):
            bad.append("opt_adam=False")
        if bad:
            raise NotImplementedError("fused mapping loop: unsupported config " + ", ".join(bad) +
                                      " (none of the shipped configs enables these)")

    def mapping(self, iter_count, index_seq: torch.Tensor = None):
        """Run `iter_count` (+ adaptive offset) training iterations on the local map
        (utils/mapper.py:620-862).  `index_seq` [iters, bs] optionally teacher-forces the batches."""
        lib = _lib.load()
        self._check_fused_config()
        cfg, nm = self.config, self.neural_points
        pipeline = int(_lib.PIPELINE if getattr(self, "pipeline", None) is None else self.pipeline)
        if not cfg.weighted_first and pipeline != 1:
            # per-neighbour decoding is fused on the hoisted schedule: csrc/train_wf0.hip (numerical or no eikonal term),
            # k_train_analytic_wf0 (analytic term: the backward through six decoder evaluations per sample in closed form)
            return self._mapping_unfused(iter_count, index_seq)
        iter_count = max(1, iter_count + self.adaptive_iter_offset)
        dist = _dist()
        world = dist.get_world_size() if dist else 1
        rank = dist.get_rank() if dist else 0
        bs_global = int(cfg.bs)
        if bs_global % world != 0:
            raise ValueError(f"batch size {bs_global} must be divisible by the world size {world}")
        bs_local = bs_global // world
        theta = nm._parameters["local_geo_features"]  # (nn.Module.__getattr__ costs a microsecond; `.data` allocates a view)
        _lib.require_cuda(theta, "local_geo_features", torch.float32)
        dev = theta.device
        n_feat = theta.numel()
        bufs = None
        if index_seq is None:
            if self.pool_sample_count <= 0:
                raise RuntimeError("mapping(): the sample pool is empty")
            bufs, index_seq = self._prepare_call(iter_count, bs_global, n_feat // _lib.F, dev, lib,
                                                 rank * bs_local if dist else 0, bs_local if dist else 0)
        else:
            iter_count = index_seq.shape[0]
            index_seq = _lib.require_cuda(index_seq.to(torch.int64).contiguous(), "index_seq", torch.int64)
        batch_offset = rank * bs_local

        W1, b1, W2, b2 = self.geo_mlp.flat_params()
        # freeze_model flips requires_grad on the decoder's children (utils/tools.py:314-317)
        train_decoder = all(p.requires_grad for p in (W1, b1, W2, b2))
        eik_mode = 0
        if cfg.ekional_loss_on and cfg.weight_e > 0:
            eik_mode = 1 if cfg.numerical_grad else 2
        # config.proj_correction_on (utils/mapper.py:57-69, 695-696, 712-714): the labels are scaled by |cos(g, x - origin)| with
        # the AUTOGRAD gradient g of every sample -- `require_gradient` wins over `numerical_grad`, so the eikonal term (if on)
        # runs on that g over the whole batch too: the analytic iteration, with the eikonal weight 0 when the term is off
        proj_corr = bool(getattr(cfg, "proj_correction_on", False))
        consistency = bool(getattr(cfg, "consistency_loss_on", False))  # utils/mapper.py:716-741: needs the autograd g as well
        weight_e = float(cfg.weight_e) if eik_mode else 0.0
        if proj_corr or consistency:
            if pipeline != 1:
                raise NotImplementedError("fused mapping loop: proj_correction_on / consistency_loss_on run on the hoisted schedule")
            eik_mode = 2
        decim = int(cfg.gradient_decimation) if eik_mode == 1 else 1
        n_eik_global = (bs_global + decim - 1) // decim

        # fused gradient buffer [decoder 833 | pad | (M+1) accumulation rows of 16 floats: 8 gradients, certainty
        # increment, 7 unused] (include/clid_native.h CLID_GRAD_ROW16) + Adam state + per-iteration losses: ONE cached
        # allocation, zeroed by one fill per call (the optimiser state restarts every call, utils/mapper.py:634)
        gstride = _lib.GRAD_ROW16
        grad, m, v, m_mlp, v_mlp, losses = bufs if bufs is not None else self._loop_buffers(n_feat // _lib.F, iter_count, dev)
        wsz = self.__dict__.setdefault("_ws_need", {})
        need = wsz.get((bs_local, decim, eik_mode))
        if need is None:
            need = wsz[(bs_local, decim, eik_mode)] = int(lib.clid_train_workspace_floats(bs_local, decim, eik_mode))
        if getattr(self, "_ws", None) is None or self._ws.numel() < need or self._ws.device != dev:
            self._ws = torch.empty(need, device=dev, dtype=torch.float32)

        view, keep = nm._map_view(True)
        pool_coord = _lib.require_cuda(self.global_coord_pool, "global_coord_pool", torch.float32)
        pool_label = _lib.require_cuda(self.sdf_label_pool, "sdf_label_pool", torch.float32)
        pool_ts = _lib.require_cuda(self.time_pool, "time_pool", torch.int32)
        pool_w = _lib.require_cuda(self.weight_pool, "weight_pool", torch.float32)

        ta = _lib.TrainArgs()
        if self.ba_done_flag:
            # utils/mapper.py:646-658: the poses of old frames moved (bundle adjustment) and global_coord_pool is stale until the
            # next process_frame re-projects it: get_batch hands out the sensor-frame coordinates and every sample is moved by
            # the pose of ITS frame -- here inside the search launch's gather (clid_train_args.pool_pose)
            pool_coord = _lib.require_cuda(self.coord_pool, "coord_pool", torch.float32)
            poses = self.used_poses
            if poses is None or poses.dim() != 3 or pipeline != 1:
                raise NotImplementedError("fused mapping loop: ba_done_flag needs used_poses [frames, 4, 4] and the hoisted schedule")
            pose34 = self._pose_slab(poses, dev)  # (`.to(points)` of utils/tools.py:624-625)
            keep = (keep, pose34)
            ta.pool_pose, ta.n_pose = pose34.data_ptr(), int(pose34.shape[0])
        ta.pool_coord, ta.pool_label, ta.pool_ts, ta.pool_weight = (
            pool_coord.data_ptr(), pool_label.data_ptr(), pool_ts.data_ptr(), pool_w.data_ptr())
        ta.bs, ta.decimation, ta.batch_offset = bs_local, decim, batch_offset
        ta.fd_eps = float(cfg.voxel_size_m * cfg.num_grad_step_ratio)
        ta.inv_n_main, ta.inv_n_eik = 1.0 / bs_global, 1.0 / n_eik_global
        ta.sigma, ta.weight_e = float(self.sdf_scale), (weight_e if (proj_corr or consistency) else float(cfg.weight_e))
        if proj_corr:
            poses = self.used_poses
            if poses is None or poses.dim() != 3:
                raise NotImplementedError("fused mapping loop: proj_correction_on needs used_poses [frames, 4, 4] (the frames' origins)")
            fpose = self._pose_slab(poses, dev)
            keep = (keep, fpose)
            ta.proj_correction, ta.frame_pose, ta.n_frame_pose = 1, fpose.data_ptr(), int(fpose.shape[0])
        ta.loss_weight_on, ta.eikonal_mode, ta.train_decoder = int(bool(cfg.loss_weight_on)), eik_mode, int(train_decoder)
        ta.main_loss_type = self.MAIN_LOSS_TYPES[cfg.main_loss_type]
        if ta.main_loss_type in (1, 2):
            ta.loss_weight_on = 1  # sdf_diff_loss applies the sample weight whatever config.loss_weight_on says (utils/loss.py:9-17)
        ta.W1, ta.b1, ta.W2, ta.b2 = W1.data_ptr(), b1.data_ptr(), W2.data_ptr(), b2.data_ptr()
        ta.sdf_scale = float(self.geo_mlp.sdf_scale)
        ta.grad, ta.ws = grad.data_ptr(), self._ws.data_ptr()
        ta.defer_reduce = 0 if dist else 1
        ta.debug_flags = int(os.environ.get('CLID_DEBUG_FLAGS', '0'))
        ta.grad_stride = gstride
        # per-call switches: ONE place decides the schedule and the kernel for Python and C alike
        ta.decode_variant = int(_lib.DECODE_VARIANT if getattr(self, "decode_variant", None) is None else self.decode_variant)
        ta.pipeline = pipeline
        ta.decode_each_neighbour = 0 if cfg.weighted_first else 1
        sdf_dbg = getattr(self, "_sdf_dbg", None)   # test aid: SDF per record slot (tile kernels)
        ta.sdf_dbg = None if sdf_dbg is None else sdf_dbg.data_ptr()
        ta.prof = getattr(self, "_prof", None)      # measurement aid: clid_profile_create() object
        if not dist:  # searches of iterations >= 1 beside the decode -> Adam chain (include/clid_native.h clid_train_args.sched)
            ta.sched, ta.side_group, ta.side_blocks = _lib.sched(dev)
        add_to = getattr(cfg, "ekional_add_to", "all")
        if eik_mode and add_to in ("surface", "freespace"):
            # utils/mapper.py:779-789: the eikonal mean over the decimated samples near / away from the surface only.  The
            # subset's size is data: counted per iteration by the search launch, used by the tile decode kernels and Adam
            if (eik_mode != 1 or dist or not cfg.weighted_first or ta.pipeline != 1
                    or not lib.clid_train_decode_kernel(C.byref(view), C.byref(ta)) > 0):
                raise NotImplementedError(f"fused mapping loop: ekional_add_to={add_to} runs on the tile decode kernels with the "
                                          "numerical eikonal term on one GPU (no shipped config sets it)")
            inv = getattr(self, "_eik_inv_n", None)
            if inv is None or inv.device != dev:
                inv = self._eik_inv_n = torch.zeros(64, device=dev, dtype=torch.float32)
            ta.eik_mask = 1 if add_to == "surface" else 2
            ta.eik_mask_range, ta.eik_inv_n = float(cfg.surface_sample_range_m), inv.data_ptr()
        hoist = ta.pipeline == 1  # (the analytic-eikonal iteration reads the hoisted search's records too)
        tile = hoist and lib.clid_train_decode_kernel(C.byref(view), C.byref(ta)) > 0

        aa = _lib.AdamArgs()
        aa.feat, aa.grad, aa.m, aa.v = theta.data_ptr(), grad.data_ptr(), m.data_ptr(), v.data_ptr()
        aa.W1, aa.b1, aa.W2, aa.b2 = ta.W1, ta.b1, ta.W2, ta.b2
        aa.m_mlp, aa.v_mlp = m_mlp.data_ptr(), v_mlp.data_ptr()
        aa.n_feat = n_feat
        aa.lr, aa.beta1, aa.beta2, aa.eps = float(cfg.lr), 0.9, 0.99, float(cfg.adam_eps)
        aa.weight_decay = float(cfg.weight_decay)
        aa.train_decoder = int(train_decoder)
        aa.grad_stride = gstride
        aa.cert, aa.n_cert = nm.local_point_certainties.data_ptr(), int(nm.local_point_certainties.shape[0])

        # Touched-row bookkeeping (include/clid_native.h clid_train_args.touch_ws): the hoisted searches flag every map
        # row every iteration will touch.  Single GPU: Adam visits only rows touched so far in this call (exact: the
        # optimiser restarts per call, utils/mapper.py:634) -- pays on local maps much larger than a batch's footprint.
        # Sharded: each iteration all-reduces [848 | 9 floats per row IT touches] instead of 64 bytes per row of the
        # local map.  CLID_SPARSE = 0 / 1 / auto.
        M_local = n_feat // _lib.F - 1
        mode = os.environ.get("CLID_SPARSE", "auto")
        if dist and getattr(self, "exchange_mode", None) in ("dense", "compact"):
            mode = "1" if self.exchange_mode == "compact" else "0"
        transport = getattr(self, "exchange_transport", None)
        want_p2p = bool(dist) and _lib.p2p_likely(dist, None if transport is None else transport == "p2p")
        floor = self.SPARSE_MIN_ROWS if not dist else (0 if want_p2p else self.SPARSE_MIN_ROWS_DIST)
        want = mode == "1" or (mode != "0" and M_local >= floor)
        use_touch = bool(want and tile and float(cfg.weight_decay) == 0.0)
        cbuf = None
        if use_touch:
            chunk = int(lib.clid_train_chunk_iters(C.byref(ta)))
            tw = getattr(self, "_touch_ws", None)  # (buffer, row capacity, chunk, device): laid out for a capacity, so
            if tw is None or tw[1] < M_local or tw[2] != chunk or tw[3] != str(dev):  # it stays put while the map grows
                m_cap = int(M_local * 1.25) + 1024
                tw = self._touch_ws = (torch.zeros(int(lib.clid_touch_workspace_bytes(m_cap, chunk)), device=dev, dtype=torch.uint8),
                                       m_cap, chunk, str(dev))
            ta.touch_ws, ta.touch_stride = tw[0].data_ptr(), int(lib.clid_touch_stride(tw[1]))
            if dist:
                n_c = _lib.GRAD_FEAT_OFFSET16 + 9 * (M_local + 1) + 16
                cbuf = getattr(self, "_cbuf", None)
                if cbuf is None or cbuf.numel() < n_c or cbuf.device != dev:
                    cbuf = self._cbuf = torch.zeros(n_c, device=dev, dtype=torch.float32)
                ta.cbuf = cbuf.data_ptr()
                # small local maps: the ranks' batches reach (nearly) every row every iteration -- all rows on every list,
                # no flag exchange and no read-back per chunk (CLID_TOUCH_ALL = 0 / 1 / auto)
                every = os.environ.get("CLID_TOUCH_ALL", "auto")
                ta.touch_all = int(every == "1" or (every != "0" and M_local < self.SPARSE_MIN_ROWS_DIST))

        if dist and cbuf is None and tile and not ta.decode_each_neighbour and eik_mode != 2:
            # dense exchange: decode -> all-reduce -> Adam (the tile kernels add their block sums into copies inside `grad`)
            rows_end = _lib.GRAD_FEAT_OFFSET16 + (n_feat // _lib.F) * _lib.GRAD_ROW16
            assert grad.numel() == rows_end + self.DEC_COPIES * 848
            ta.dec_copies, ta.n_dec_copies, ta.dec_ranks = grad.data_ptr() + 4 * rows_end, self.DEC_COPIES, world
        stream = _lib.stream()
        idx_base, row_bytes = index_seq.data_ptr(), bs_global * 8
        loss_base = losses.data_ptr()
        if getattr(self, "_grad_probe", False):
            return self._probe_gradients(lib, view, keep, ta, grad, losses, idx_base, bs_global, n_feat // _lib.F, dev, stream)
        cert_in_rows = True
        saved = None
        try:
            if not dist:
                # single GPU: the whole loop is enqueued by one C call (hoisted searches + 2 launches per iteration)
                ta.index, ta.loss_out = idx_base, loss_base
                if consistency:
                    self._consistency_loop(lib, view, ta, aa, iter_count, index_seq, bs_global, losses, dev, stream)
                else:
                    _lib.check(lib.clid_mapping_run(C.byref(view), C.byref(ta), C.byref(aa), iter_count, idx_base, bs_global,
                                                    loss_base, stream), "clid_mapping_run")
            else:
                # the neighbour searches do not depend on the training state: one launch per chunk of iterations
                # resolves this rank's shard of every batch, then decode/backward -> all-reduce -> Adam per iteration.
                # With the tile kernels the certainty increments ride in the all-reduced accumulation rows (every rank's
                # Adam launch applies the global sum), and so they do in the analytic-eikonal and weighted_first: False iterations (their pairs leave as
                # whole rows); the other 16-lane kernels add this rank's share to the array directly
                cert_in_rows = tile or eik_mode == 2 or bool(ta.decode_each_neighbour)
                cert0 = None if cert_in_rows else nm.local_point_certainties.clone()
                comm = _lib.rccl_comm(dist)
                if cbuf is not None and hoist and want_p2p:
                    # compact exchange over peer-mapped buffers when asked for and every rank could set them up
                    # (csrc/p2p.hip: one launch per rank instead of an RCCL ring); RCCL / torch.distributed otherwise
                    ta.p2p = _lib.p2p_exchange(dist, 4 * cbuf.numel())
                    if ta.p2p:
                        # a flag wait that gives up leaves garbage sums behind Adam steps already taken: keep what the
                        # loop mutates, so the call can be repeated over RCCL (four small device copies per call)
                        saved = self._save_trained_state(nm, (W1, b1, W2, b2))
                shard_base = idx_base + batch_offset * 8
                if comm is not None:
                    # RCCL behind the C ABI: the whole sharded loop is ONE host call, the all-reduce sits on the launch
                    # stream between the partial reduction and Adam (csrc/train.hip clid_mapping_run_dist)
                    ta.index, ta.loss_out = shard_base, loss_base
                    moved = C.c_int64(0)
                    _lib.check(lib.clid_mapping_run_dist(C.byref(view), C.byref(ta), C.byref(aa), iter_count, shard_base,
                                                         bs_global, loss_base, comm, grad.numel(), C.byref(moved), stream),
                               "clid_mapping_run_dist")
                    merged = True  # losses (SUM) and update stamps (MAX) were merged by the call
                    moved = int(moved.value)
                else:
                    merged = False
                    moved = self._mapping_loop_torch_dist(lib, dist, view, ta, aa, grad, cbuf, iter_count, shard_base, row_bytes,
                                                          loss_base, hoist, bs_local, batch_offset, decim, eik_mode, dev, stream)
                self.last_exchange = {"mode": "compact" if (use_touch and hoist) else "dense", "iters": iter_count,
                                      "every_row": bool(ta.touch_all) and use_touch and hoist,
                                      "transport": "peer-mapped" if ta.p2p else ("rccl" if comm is not None else "torch.distributed"),
                                      "floats": moved, "bytes_per_iter": 4.0 * moved / max(iter_count, 1),
                                      "dense_bytes_per_iter": 4.0 * grad.numel(), "rows": M_local + 1}
        except _lib.P2pTimeout as exc:
            # agreed across the ranks (clid_p2p_agree / the MAX below): EVERY rank is here.  Restore, rule the transport
            # out for this process, repeat the same batches over RCCL.
            self._touch_ws = None
            if saved is None:
                raise
            self._restore_trained_state(nm, (W1, b1, W2, b2), saved)
            _lib.p2p_disable()
            self.p2p_fallbacks += 1
            self.last_p2p_error = str(exc)
            del keep
            return self.mapping(0, index_seq=index_seq)
        except Exception:
            self._touch_ws = None  # its flags may be half-written: the next call starts from a zeroed workspace
            raise
        self.total_iter += iter_count
        if dist:
            # merge the replicas' side effects once per call (not read inside the loop's loss)
            if not cert_in_rows:
                inc = nm.local_point_certainties - cert0
                dist.all_reduce(inc)
                nm.local_point_certainties.copy_(cert0 + inc)
            if not merged:
                dist.all_reduce(nm.local_point_ts_update, op=dist.ReduceOp.MAX)
                dist.all_reduce(losses)
            self._mapping_calls = getattr(self, "_mapping_calls", 0) + 1
            # every call: the sizes the exchange depends on (cheap, asynchronous, verified at the next call's end); every
            # `replica_check_every`-th call: the content sums too
            self._queue_size_check(dist, M_local, iter_count)
            if (self._mapping_calls - 1) % max(int(getattr(self, "replica_check_every", 16)), 1) == 0:
                self._check_replicas(dist)
        self.last_losses = losses
        self._keep = (keep, index_seq, grad, m, v, m_mlp, v_mlp)
        nm.assign_local_to_global()

    def _pose_slab(self, poses: torch.Tensor, dev) -> torch.Tensor:
        """used_poses[:, :3, :] as one contiguous fp32 [frames, 12] slab on `dev` (clid_train_args.pool_pose / frame_pose), rebuilt
        only when the pose tensor changed (storage, in-place version, shape): ba_done_flag / proj_correction calls of one frame
        share it."""
        key = (poses.data_ptr(), poses._version, tuple(poses.shape), str(poses.dtype), str(dev))
        hit = self.__dict__.get("_pose_slab_cache")
        if hit is None or hit[0] != key:
            hit = self._pose_slab_cache = (key, poses[:, :3, :].to(device=dev, dtype=torch.float32).contiguous())
        return hit[1]

    def _consistency_loop(self, lib, view, ta, aa, iter_count, index_seq, bs, losses, dev, stream):
        """config.consistency_loss_on (utils/mapper.py:716-741, 770-776; "[not used]" there): per iteration a second batch of
        min(consistency_count, bs) randomly shifted copies of drawn samples is searched and both batches run the analytic iteration
        twice -- a probe that only evaluates g = d sdf / d x, `clid_consistency_couple` (the term 1 - cos(g, g_near), its value and
        dL/dg of both batches), then the backward proper with that dL/dg added.  Host-driven (seven launches + two small torch
        gathers per iteration): this branch is not on anybody's hot path.  The two extra draws of an iteration come from torch's
        generator like the reference's, or from `self._consistency_draws` = [(near_index [n_c], random_shift [bs, 3])] (tests:
        the reference's recorded draws)."""
        cfg = self.config
        n_c = min(int(cfg.consistency_count), bs)
        rng, wc = float(cfg.consistency_range), float(cfg.weight_c)
        bufs = self.__dict__.get("_cons_bufs")
        if bufs is None or bufs["key"] != (bs, n_c, str(dev)):
            f32 = dict(device=dev, dtype=torch.float32)
            bufs = self._cons_bufs = {
                "key": (bs, n_c, str(dev)),
                "rec_main": torch.empty(int(lib.clid_train_search_floats(bs, 0, 1, 2, 1)), **f32),
                "rec_near": torch.empty(int(lib.clid_train_search_floats(n_c, 0, 1, 2, 1)), **f32),
                "g_main": torch.zeros((bs, 3), **f32), "g_near": torch.zeros((n_c, 3), **f32),
                "c_main": torch.zeros((bs, 3), **f32), "c_near": torch.zeros((n_c, 3), **f32),
                "near": torch.zeros((n_c, 3), **f32), "zeros": torch.zeros(n_c, **f32),
                "arange": torch.arange(n_c, device=dev, dtype=torch.int64),
            }
        b = bufs
        tn = _lib.TrainArgs.from_buffer_copy(ta)  # the shifted copies as a batch of their own: their coordinates are its pool,
        tn.pool_coord, tn.pool_label, tn.pool_weight, tn.pool_ts = (  # labels / weights zero: no BCE term, no stamps (query_ts None)
            b["near"].data_ptr(), b["zeros"].data_ptr(), b["zeros"].data_ptr(), None)
        tn.index, tn.bs, tn.batch_offset, tn.decimation = b["arange"].data_ptr(), n_c, 0, 1
        tn.weight_e, tn.loss_weight_on, tn.proj_correction, tn.pool_pose = 0.0, 1, 0, None
        nb_main = int(lib.clid_train_partial_rows(C.byref(ta)))
        nb_near = int(lib.clid_train_partial_rows(C.byref(tn)))
        draws = getattr(self, "_consistency_draws", None)
        if draws is not None and len(draws) < iter_count:  # (checked before the first Adam step: a short list must not stop the loop midway)
            raise ValueError(f"_consistency_draws holds {len(draws)} iterations, the call runs {iter_count} "
                             "(iter_count + adaptive_iter_offset)")
        pool = self.global_coord_pool
        for it in range(iter_count):
            index = index_seq[it]
            loss_row = losses[it].data_ptr()
            coord = pool.index_select(0, index)
            if draws is not None:
                near_index = draws[it][0].to(device=dev, dtype=torch.int64)
                shift = draws[it][1].to(device=dev, dtype=torch.float32)
            else:
                near_index = torch.randint(0, bs, (n_c,), device=dev)
                shift = torch.rand_like(coord) * 2 * rng - rng
            near_index = near_index.contiguous()
            b["near"].copy_((coord + shift).index_select(0, near_index))
            ta.index, ta.loss_out, tn.loss_out = index.data_ptr(), loss_row, loss_row
            for t_, rec_, n_ in ((ta, b["rec_main"], bs), (tn, b["rec_near"], n_c)):
                t_.g_out, t_.c_extra, t_.partial_row0, t_.partial_rows_extra = None, None, 0, 0
                _lib.check(lib.clid_train_search(C.byref(view), C.byref(t_), 1, t_.index, n_, rec_.data_ptr(), stream), "clid_train_search")
            for t_, rec_, g_ in ((ta, b["rec_main"], b["g_main"]), (tn, b["rec_near"], b["g_near"])):  # the two probes
                t_.g_out = g_.data_ptr()
                _lib.check(lib.clid_train_decode(C.byref(view), C.byref(t_), rec_.data_ptr(), stream), "clid_train_decode")
                t_.g_out = None
            _lib.check(lib.clid_consistency_couple(b["g_main"].data_ptr(), b["g_near"].data_ptr(), near_index.data_ptr(), n_c, bs, wc,
                                                   b["c_main"].data_ptr(), b["c_near"].data_ptr(), loss_row, stream),
                       "clid_consistency_couple")
            ta.c_extra, ta.partial_rows_extra = b["c_main"].data_ptr(), nb_near
            tn.c_extra, tn.partial_row0 = b["c_near"].data_ptr(), nb_main
            _lib.check(lib.clid_train_decode(C.byref(view), C.byref(ta), b["rec_main"].data_ptr(), stream), "clid_train_decode")
            _lib.check(lib.clid_train_decode(C.byref(view), C.byref(tn), b["rec_near"].data_ptr(), stream), "clid_train_decode")
            aa.step = it + 1
            _lib.check(lib.clid_train_adam(C.byref(aa), C.byref(ta), stream), "clid_train_adam")
            del near_index, shift, coord  # (their launches are enqueued on this stream: the caching allocator keeps the order)
        ta.c_extra, ta.partial_rows_extra = None, 0

    def _save_trained_state(self, nm, dec_params):
        """Copies of everything a mapping() call mutates (features, decoder, certainties, update stamps) in cached
        buffers."""
        src = [nm.local_geo_features.data, nm.local_point_certainties, nm.local_point_ts_update] + [p.data for p in dec_params]
        bufs = getattr(self, "_saved_state", None)
        if bufs is None or any(b.shape != t.shape or b.device != t.device or b.dtype != t.dtype for b, t in zip(bufs, src)):
            bufs = self._saved_state = [torch.empty_like(t) for t in src]
        for b, t in zip(bufs, src):
            b.copy_(t)
        return bufs

    def _restore_trained_state(self, nm, dec_params, bufs):
        dst = [nm.local_geo_features.data, nm.local_point_certainties, nm.local_point_ts_update] + [p.data for p in dec_params]
        for b, t in zip(bufs, dst):
            t.copy_(b)

    def _probe_gradients(self, lib, view, keep, ta, grad, losses, idx_base, bs, n_rows, dev, stream):
        """Checker aid (tests, bench_sequence --check-frames; `self._grad_probe = True` then `mapping(1, index_seq=...)`):
        the gradients of ONE iteration on the current state -- search + decode through the C ABI, NO optimiser step --
        as {"theta": [rows, 8], "decoder": [833], "cert_inc": [rows], "loss": [4]}.  Gradients are linear in the per-query
        terms, so unlike parameters after several eps = 1e-15 Adam steps they can be compared entry by entry with a CPU
        evaluation of the reference's loop."""
        if ta.eikonal_mode == 2 or ta.pipeline != 1 or _dist() is not None:
            raise NotImplementedError("gradient probe: hoisted single-GPU schedule only")
        ta.defer_reduce, ta.touch_ws, ta.cbuf = 0, None, None
        ta.index, ta.loss_out = idx_base, losses.data_ptr()
        rec = torch.empty(int(lib.clid_train_search_floats(ta.bs, ta.batch_offset, ta.decimation, ta.eikonal_mode, 1)),
                          device=dev, dtype=torch.float32)
        ts_before = self.neural_points.local_point_ts_update.clone()
        sdf_slots = None
        if getattr(self, "_probe_sdf", False):  # also hand back the task records and the SDF of every record slot
            n_tasks = int(lib.clid_train_search_tasks(ta.bs, ta.batch_offset, ta.decimation, ta.eikonal_mode))
            sdf_slots = torch.zeros(n_tasks * 8, device=dev)
            ta.sdf_dbg = sdf_slots.data_ptr()
        _lib.check(lib.clid_train_search(C.byref(view), C.byref(ta), 1, idx_base, bs, rec.data_ptr(), stream), "clid_train_search")
        _lib.check(lib.clid_train_decode(C.byref(view), C.byref(ta), rec.data_ptr(), stream), "clid_train_decode")
        torch.cuda.synchronize()
        rows = grad[_lib.GRAD_FEAT_OFFSET16:_lib.GRAD_FEAT_OFFSET16 + n_rows * _lib.GRAD_ROW16].view(n_rows, _lib.GRAD_ROW16)
        out = {"theta": rows[:, :_lib.F].clone(), "decoder": grad[:_lib.MLP_PARAMS].clone(), "cert_inc": rows[:, _lib.F].clone(),
               "loss": losses[0].clone()}
        if sdf_slots is not None:
            out["records"] = rec[: sdf_slots.numel() // 8 * 192].view(-1, 48, 4).clone()
            out["sdf_slots"] = sdf_slots.view(-1, 8)
        grad.zero_()
        losses.zero_()
        self.neural_points.local_point_ts_update.copy_(ts_before)  # (the decode's only direct side effect with the tile kernels)
        del keep
        return out

    SPARSE_MIN_ROWS = 1 << 16       # single GPU: local maps from this size on run the touched-row Adam sweep
    SPARSE_MIN_ROWS_DIST = 1 << 15  # sharded: from here on the compact exchange (dense payload 64 B x rows > 2 MB)

    def _mapping_loop_torch_dist(self, lib, dist, view, ta, aa, grad, cbuf, iter_count, shard_base, row_bytes, loss_base,
                                 hoist, bs_local, batch_offset, decim, eik_mode, dev, stream):
        """The sharded loop with torch.distributed's all-reduce between two C calls per iteration: the path for
        non-RCCL backends (gloo dry runs / tests with several ranks on one GPU).  The same sequence as
        clid_mapping_run_dist (csrc/train.hip), exchange included; returns the 4-byte words this rank all-reduced."""
        moved = 0
        compact = bool(hoist and ta.touch_ws and cbuf is not None)
        ta.pipeline = 1 if hoist else 0
        if not hoist:
            ta.touch_ws, ta.cbuf = None, None
        elif not compact:
            ta.cbuf = None
        chunk = 1
        if hoist:
            chunk = int(lib.clid_train_chunk_iters(C.byref(ta)))
            per_iter = int(lib.clid_train_search_floats(bs_local, batch_offset, decim, eik_mode, 1))
            if getattr(self, "_rec", None) is None or self._rec.numel() < per_iter * chunk or self._rec.device != dev:
                self._rec = torch.empty(per_iter * chunk, device=dev, dtype=torch.float32)
        counts = (C.c_int32 * 32)()
        M_local = int(view.M)
        px = ta.p2p if compact else None
        for it in range(iter_count):
            ta.index = shard_base + it * row_bytes
            ta.loss_out = loss_base + it * 16
            ta.touch_iter = it % chunk
            if hoist:
                if it % chunk == 0:
                    n_it = min(chunk, iter_count - it)
                    _lib.check(lib.clid_train_search(C.byref(view), C.byref(ta), n_it, ta.index, row_bytes // 8,
                                                     self._rec.data_ptr(), stream), "clid_train_search")
                    if ta.touch_ws and ta.touch_all and compact:
                        self._touch_ws[0][: n_it * int(ta.touch_stride)].view(n_it, int(ta.touch_stride))[:, :M_local] = 1
                        _lib.check(lib.clid_train_touch_scan(C.byref(ta), M_local, n_it, it, None, stream), "clid_train_touch_scan")
                        for i in range(n_it):
                            counts[i] = M_local
                    elif ta.touch_ws:
                        flags = self._touch_ws[0][: n_it * int(ta.touch_stride)]
                        if px and flags.numel() + 16 <= int(lib.clid_p2p_capacity(px)):
                            _lib.check(lib.clid_p2p_allreduce_or(px, flags.data_ptr(), flags.numel(), stream), "clid_p2p_allreduce_or")
                        else:
                            dist.all_reduce(flags, op=dist.ReduceOp.MAX)  # union over the ranks of each iteration's rows
                        moved += (flags.numel() + 3) // 4
                        _lib.check(lib.clid_train_touch_scan(C.byref(ta), M_local, n_it, it, counts if compact else None, stream),
                                   "clid_train_touch_scan")
                if px:  # (after the chunk's flag exchange, which takes one turn of the two exchange buffers itself)
                    ta.cbuf = lib.clid_p2p_buffer(px)
                _lib.check(lib.clid_train_decode(C.byref(view), C.byref(ta),
                                                 self._rec.data_ptr() + (it % chunk) * per_iter * 4, stream),
                           "clid_train_decode")
            else:
                _lib.check(lib.clid_train_fwd_bwd(C.byref(view), C.byref(ta), stream), "clid_train_fwd_bwd")
            if compact:
                n = _lib.GRAD_FEAT_OFFSET16 + 9 * int(counts[it % chunk])
                if px:
                    _lib.check(lib.clid_p2p_allreduce(px, n, stream), "clid_p2p_allreduce")
                else:
                    dist.all_reduce(cbuf[:n])
                moved += n
            else:
                dist.all_reduce(grad)
                moved += grad.numel()
            aa.step = it + 1
            _lib.check(lib.clid_train_adam(C.byref(aa), C.byref(ta), stream), "clid_train_adam")
        if px and iter_count > 0:
            # a flag wait that gave up on ANY rank invalidates the call on EVERY rank: agree (MAX) before anybody decides
            rc = int(lib.clid_p2p_status(px, stream))
            if rc not in (0, _lib.E_P2P_TIMEOUT):
                _lib.check(rc, "clid_p2p_status")
            bad = torch.tensor([1 if rc else 0], dtype=torch.int32, device=dev if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(bad, op=dist.ReduceOp.MAX)
            if int(bad.item()):
                raise _lib.P2pTimeout("clid_p2p: a rank of the group gave up waiting at an exchange; the call's sums are invalid")
        return moved

    def _queue_size_check(self, dist, M_local: int, iter_count: int):
        """with each rank drawing only its own shard, nothing cross-checked the replicas between two full checks.
        Every sharded call now MIN-all-reduces [x, -x] over (local map size, pool size, new-sample count, iterations, draw
        counter) -- one tiny asynchronous collective -- and the PREVIOUS call's result is examined here, when it has long
        arrived (no stall): different sizes mean different row lists and exchange lengths, i.e. garbage sums."""
        wire = self.neural_points.local_geo_features.device if dist.get_backend() == "nccl" else torch.device("cpu")
        prev = self.__dict__.pop("_size_check", None)
        if prev is not None:
            work, t, mine = prev
            work.wait()
            got = t.tolist()
            n = len(mine)
            if got[:n] != mine or [-v for v in got[n:]] != mine:
                raise RuntimeError(f"data-parallel replicas diverged: this rank had (M_local, pool, new samples, iterations, draws) = {mine}, "
                                   f"the group's min / max are {got[:n]} / {[-v for v in got[n:]]}")
        mine = [int(M_local), int(self.pool_sample_count), int(0 if self.new_idx is None else self.new_idx.shape[0]), int(iter_count),
                int(getattr(self, "_draw_calls", 0))]
        t = torch.tensor(mine + [-v for v in mine], dtype=torch.int64, device=wire)
        self._size_check = (dist.all_reduce(t, op=dist.ReduceOp.MIN, async_op=True), t, mine)

    def _check_replicas(self, dist):
        """Data parallelism here relies on every rank holding a bit-identical replica of the map and the pool (same
        seeds, same frames).  A consistency check on the first and then every `replica_check_every`-th (default 16)
        `mapping()` call -- it costs four device reductions, a collective and a host synchronisation --: pool size, local
        map size, the pool's labels, the features and the certainties must agree on all ranks (MIN == MAX), else the
        gradients that were just summed belong to different samples / rows."""
        nm = self.neural_points
        dev = nm.local_geo_features.device
        sig = torch.stack((
            torch.tensor([float(self.pool_sample_count), float(nm.local_count()), float(nm.count()),
                          float(0 if self.new_idx is None else self.new_idx.shape[0])], device=dev, dtype=torch.float64).sum(),
            self.sdf_label_pool.sum(dtype=torch.float64),                 # pool CONTENT (the samplers' draws)
            nm.local_geo_features.data.sum(dtype=torch.float64),          # features after the identical Adam steps
            nm.local_point_certainties.sum(dtype=torch.float64)))         # side effects merged by the call
        both = torch.cat((sig, -sig))
        dist.all_reduce(both, op=dist.ReduceOp.MIN)  # one collective: min(x) and -max(x)
        lo, hi = both[:4], -both[4:]
        if not torch.equal(lo, hi):
            raise RuntimeError(
                f"data-parallel replicas diverged ([counts, pool label sum, feature sum, certainty sum] min {lo.tolist()} max {hi.tolist()}): "
                "every rank must process the same frames with the same seeds (clid_slam_amd seeds its own generators from "
                "config.seed; see INTEGRATION.md)")

    def _mapping_unfused(self, iter_count, index_seq=None):
        """`weighted_first: False` (decode every neighbour, blend the SDFs; utils/mapper.py:679-680): the
        reference's op sequence on the HIP-backed autograd ops (query_feature / Decoder.sdf / sdf_bce_loss
        kernels) with torch's Adam.  Single GPU, numerical or no eikonal term."""
        from .loss import sdf_bce_loss
        from .tools import setup_optimizer

        cfg, nm = self.config, self.neural_points
        if _dist() is not None:
            raise NotImplementedError("weighted_first=False is not sharded across GPUs")
        if cfg.ekional_loss_on and cfg.weight_e > 0 and not cfg.numerical_grad:
            raise NotImplementedError("weighted_first=False with the analytic eikonal term needs double backward")
        iter_count = max(1, iter_count + self.adaptive_iter_offset)
        if index_seq is None:
            index_seq = self._draw_index(iter_count, int(cfg.bs))
        iter_count = index_seq.shape[0]
        opt = setup_optimizer(cfg, list(nm.parameters()), list(self.geo_mlp.parameters()))
        losses = torch.zeros((iter_count, 4), device=self.global_coord_pool.device)
        for it in range(iter_count):
            index = index_seq[it].to(torch.int64)
            coord, label = self.global_coord_pool[index], self.sdf_label_pool[index]
            ts, weight = self.time_pool[index], self.weight_pool[index].abs()
            feat, _, w_knn, _, _ = nm.query_feature(coord, ts)
            sdf_pred = torch.sum(self.geo_mlp.sdf(feat) * w_knn, dim=1).squeeze(1)
            loss = sdf_bce_loss(sdf_pred, label, self.sdf_scale, weight, cfg.loss_weight_on)
            losses[it, 1] = loss.detach()
            if cfg.ekional_loss_on and cfg.weight_e > 0:
                d = cfg.gradient_decimation
                g = self.get_numerical_gradient(coord[::d], sdf_pred[::d], cfg.voxel_size_m * cfg.num_grad_step_ratio)
                eik = ((g.norm(2, dim=-1) - 1.0) ** 2).mean()
                losses[it, 2] = eik.detach()
                loss = loss + cfg.weight_e * eik
            losses[it, 0] = loss.detach()
            opt.zero_grad(set_to_none=True)
            loss.backward()
            opt.step()
        self.total_iter += iter_count
        self.last_losses = losses
        nm.assign_local_to_global()

    # ------------------------------------------------------------------ a5 / a6
    def sdf(self, x, get_std=False):
        """utils/mapper.py:968-982: default-argument query (training_mode=True) + decoder."""
        geo_feature, _, weight_knn, _, _ = self.neural_points.query_feature(x)
        sdf_pred = self.geo_mlp.sdf(geo_feature)
        sdf_std = None
        if not self.config.weighted_first:
            sdf_pred_mean = torch.sum(sdf_pred * weight_knn, dim=1)
            if get_std:
                sdf_var = torch.sum(weight_knn * (sdf_pred - sdf_pred_mean.unsqueeze(-1)) ** 2, dim=1)
                sdf_std = torch.sqrt(sdf_var).squeeze(1)
            sdf_pred = sdf_pred_mean.squeeze(1)
        return sdf_pred, sdf_std

    def get_numerical_gradient(self, x, sdf_x=None, eps=0.02, two_side=True):
        """utils/mapper.py:985-1034: central (or forward) differences through ONE `sdf` call."""
        n = x.shape[0]
        shifts = []
        for a in range(3):
            e = torch.zeros(3, dtype=x.dtype, device=x.device)
            e[a] = eps
            shifts.append(e)
        if two_side:
            pts = torch.cat([x + s * sg for s in shifts for sg in (1.0, -1.0)], dim=0)
            s = self.sdf(pts)[0].unsqueeze(-1)
            cols = [(s[2 * a * n:(2 * a + 1) * n] - s[(2 * a + 1) * n:(2 * a + 2) * n]) / (2 * eps) for a in range(3)]
        else:
            pts = torch.cat([x + s for s in shifts], dim=0)
            s = self.sdf(pts)[0].unsqueeze(-1)
            base = sdf_x.unsqueeze(-1)
            cols = [(s[a * n:(a + 1) * n] - base) / eps for a in range(3)]
        return torch.cat(cols, dim=1)

    # ------------------------------------------------------------------ pool helpers / out of scope
    def set_pool(self, global_coord, sdf_label, weight, time, new_idx=None):
        """Fill the training pool directly (what `process_frame` does after sampling,
        utils/mapper.py:297-333), for callers that generate samples themselves."""
        dev = self.device
        self.global_coord_pool = global_coord.to(dev, torch.float32).contiguous()
        self.coord_pool = self.global_coord_pool
        self.sdf_label_pool = sdf_label.to(dev, torch.float32).contiguous()
        self.weight_pool = weight.to(dev, torch.float32).contiguous()
        self.time_pool = time.to(dev, torch.int32).contiguous()
        self.pool_sample_count = self.global_coord_pool.shape[0]
        self.new_idx = None if new_idx is None else new_idx.to(dev, torch.int64).contiguous()

    def free_pool(self):
        self.coord_pool = self.global_coord_pool = self.weight_pool = None
        self.sdf_label_pool = self.color_pool = self.sem_label_pool = self.normal_label_pool = None
        self.time_pool = None

    # ------------------------------------------------------------------ N2: per-frame glue in front of the loop
    def determine_used_pose(self):
        """utils/mapper.py:138-157."""
        cur = self.dataset.processed_frame
        cfg = self.config
        if getattr(cfg, "pgo_on", False):
            poses = self.dataset.pgo_poses
        elif getattr(cfg, "track_on", False):
            poses = self.dataset.odom_poses
        elif getattr(self.dataset, "gt_pose_provided", False):
            poses = self.dataset.gt_poses
        else:
            return
        src = poses[: cur + 1]
        if isinstance(src, np.ndarray):
            # uploaded when somebody reads `used_poses` (the un-fused pool path): a pageable host-to-device copy per frame
            # synchronises the stream, and nothing on the fused path reads the tensor.  Copied now, as the reference's
            # torch.tensor() copies now (odometry poses are updated in place later).
            self._used_poses, self._used_poses_src = None, src.copy()
        else:
            self.used_poses = torch.as_tensor(src, device=self.device, dtype=torch.float64)

    @property
    def used_poses(self):
        src = self.__dict__.get("_used_poses_src")
        if src is not None:
            self._used_poses, self._used_poses_src = torch.as_tensor(src, device=self.device, dtype=torch.float64), None
        return self.__dict__.get("_used_poses")

    @used_poses.setter
    def used_poses(self, value):
        self._used_poses, self._used_poses_src = value, None

    def dynamic_filter(self, points_torch, type_2_on: bool = True):
        """utils/mapper.py:99-136: static = not (certainly free space), and the SDF gradient looks sane."""
        cfg = self.config
        sdf_pred, grad, _, certainty = self.neural_points.query_sdf_and_gradient(self.geo_mlp, points_torch)
        static = (certainty < cfg.dynamic_certainty_thre) | (sdf_pred < cfg.dynamic_sdf_ratio_thre * cfg.voxel_size_m)
        if type_2_on:
            static = static & ((grad.norm(dim=-1) > getattr(cfg, "dynamic_min_grad_norm_thre", 0.3)) | (certainty < cfg.dynamic_certainty_thre))
        return static

    def process_frame(self, point_cloud_torch, frame_label_torch, cur_pose_torch, frame_id: int,
                      filter_dynamic: bool = False):
        """utils/mapper.py:159-470: raw-point map update -> sample + label this scan -> grow the neural-point
        map -> append to / filter the training pool -> pick the newly observed samples and the adaptive
        iteration offset.  Sampling is one HIP launch (data_sampler.py); the rest is per-frame bookkeeping."""
        from .tools import transform_torch

        cfg, nm = self.config, self.neural_points
        if getattr(self, "sampler", None) is None:  # e.g. the reference's Mapper subclassed over this method
            from .data_sampler import DataSampler

            self.sampler = DataSampler(cfg)
        origin = cur_pose_torch[:3, 3]
        orientation = cur_pose_torch[:3, :3]
        cur_pose_torch = _lib.small_to_host(cur_pose_torch)  # one read-back: the kernels take the 12 pose numbers by value
        pts = point_cloud_torch[:, :3]
        use_pin = bool(getattr(cfg, "use_pin_mapper", False))
        if not use_pin and not filter_dynamic and pts.is_cuda and hasattr(self.sampler, "predraw"):
            # the sampler's draws do not depend on the raw-point map: enqueued now, they are generated while the host waits
            # for the map update's round trips (same generator, same order as at their old place: nothing in between draws)
            self.sampler.predraw(pts.shape[0], pts.device)
        async_vox = os.environ.get("CLID_ASYNC_VOXEL", "1") != "0"  # voxel passes whose counts nobody reads on their own
        if not use_pin:  # :178-183
            self.local_point_cloud_map._sensor_pos_host = (origin, tuple(float(v) for v in cur_pose_torch[:3, 3].tolist()))
            # the raw-point map's new size is needed by nobody before the frame's last read-back: it lands in the frame's
            # count block and is read there (the sampler's kernels take the upper bound meanwhile)
            # ... only where the fused sampler + compaction path, which reads the size on the device, is certain to follow
            # (with the dynamic filter or the un-fused sampler a row-wise consumer would have seen the upper bound's
            # uninitialised tail)
            fused_follows = (not filter_dynamic and pts.is_cuda and hasattr(self.sampler, "_run") and getattr(cfg, "from_sample_points", True)
                             and not getattr(cfg, "from_all_samples", False) and os.environ.get("CLID_FUSED_COMPACT", "1") != "0")
            self.local_point_cloud_map._defer_counts = (
                self._frame_count_block(pts.device)[4:6] if fused_follows and os.environ.get("CLID_DEFER_CLOUD_COUNT", "1") != "0" else None)
            if self.local_point_cloud_map._defer_counts is not None and async_vox:
                self.local_point_cloud_map._defer_vox = self._frame_counts[8:10]
            self.local_point_cloud_map.update_map(origin, transform_torch(pts, cur_pose_torch))
        self.static_mask = torch.ones(pts.shape[0], dtype=torch.bool, device=pts.device)
        if filter_dynamic:  # :189-204
            nm.reset_local_map(origin, orientation, frame_id)
            self.static_mask = self.dynamic_filter(transform_torch(pts, cur_pose_torch))
            pts = pts[self.static_mask]
            if frame_label_torch is not None:
                frame_label_torch = frame_label_torch[self.static_mask]
        self.dataset.static_mask = self.static_mask
        color = None
        if getattr(cfg, "color_on", False):
            color = point_cloud_torch[:, 3:]
            color = color[self.static_mask] if filter_dynamic else color

        normal_label = sem_label = color_label = None
        gcoord = None
        defer_cmp = False
        fused_compact = (not use_pin and pts.is_cuda and hasattr(self.sampler, "_run") and getattr(cfg, "from_sample_points", True)
                         and not getattr(cfg, "from_all_samples", False) and os.environ.get("CLID_FUSED_COMPACT", "1") != "0")
        if use_pin:  # :224-239
            coord, sdf_label, normal_label, sem_label, color_label, weight = self.sampler.sample_pin(
                pts, None, frame_label_torch, color)
        elif fused_compact:
            # :240-283 + :297-310 in one enqueue: compaction of the sampler's rows, frame stamps, world-frame coordinates for
            # the pool and the near-surface world-frame subset that grows the map; ONE read-back for the two counts
            # Where the pool maintenance will run on the side stream (below), nothing needs the compaction's two counts before
            # the map growth's voxel count is read: they stay on the device for the launches in between and are read then.
            defer_cmp = (self.sem_label_pool is None and self.color_pool is None and self.normal_label_pool is None
                         and not self.ba_done_flag and (frame_id + 1) % getattr(cfg, "pool_filter_freq", 1) == 0
                         and cur_pose_torch.dtype == torch.float64 and self.coord_pool.shape[0] + 8 * pts.shape[0] < (1 << 31)
                         and os.environ.get("CLID_FUSED_POOL", "1") != "0" and os.environ.get("CLID_POOL_OVERLAP", "1") != "0"
                         and os.environ.get("CLID_DEFER_COMPACT_COUNT", "1") != "0" and pts.shape[0] > 0)
            async_upd = defer_cmp and async_vox and nm.update_is_fused()
            coord, gcoord, sdf_label, weight, stamp, update_points = self._sample_compact_fused(
                pts, cur_pose_torch, frame_id, defer_counts=defer_cmp, counts=nm.update_counts(pts.device)[5:7] if async_upd else None)
        else:  # :240-245, the region-specific SDF estimation
            coord, sdf_label, weight = self.sampler.sample(pts, self.local_point_cloud_map, cur_pose_torch)
        n_cur = coord.shape[0]
        if not fused_compact:
            stamp = torch.full((n_cur,), frame_id, dtype=torch.int, device=coord.device)
        self.cur_sample_count = n_cur
        self.pool_sample_count = self.sdf_label_pool.shape[0]

        # grow the neural-point map from the samples closest to the surface (:257-283)
        if fused_compact:
            pass
        elif getattr(cfg, "from_sample_points", True):
            if getattr(cfg, "from_all_samples", False):
                update_points = coord  # (sensor frame, as the reference passes it)
            else:
                near = torch.abs(sdf_label) < cfg.surface_sample_range_m * getattr(cfg, "map_surface_ratio", 0.5)
                update_points = transform_torch(coord[near, :], cur_pose_torch)
        else:
            update_points = transform_torch(pts, cur_pose_torch)
        if getattr(cfg, "prune_map_on", False) and (frame_id + 1) % cfg.prune_freq_frame == 0:
            if nm.prune_map(cfg.max_prune_certainty):
                nm.recreate_hash(None, None, True, True, frame_id)
        nm._sensor_pos_host = (origin, tuple(float(v) for v in cur_pose_torch[:3, 3].tolist()))  # spares reset_local_map a read-back
        fused_pool = (coord.is_cuda and sem_label is None and color_label is None and normal_label is None
                      and self.sem_label_pool is None and self.color_pool is None and self.normal_label_pool is None
                      and not self.ba_done_flag and (frame_id + 1) % getattr(cfg, "pool_filter_freq", 1) == 0
                      and cur_pose_torch.dtype == torch.float64  # the window test is float64 by type promotion (:346-349)
                      and self.coord_pool.shape[0] + n_cur < (1 << 31) and os.environ.get("CLID_FUSED_POOL", "1") != "0")
        # The pool maintenance (:297-392) and the map growth (:257-283) touch disjoint state: the pool's launches (the
        # frame's largest, bandwidth-bound at a full pool) go to a side stream and run under NeuralPoints.update, whose
        # launches are small and separated by its read-backs; joined before anything reads the pool.
        overlap = fused_pool and gcoord is not None and os.environ.get("CLID_POOL_OVERLAP", "1") != "0"
        vox_idx = None
        if overlap:
            # the map growth starts with a voxel down-sampling whose kernels starve next to the pool's 230-us five-array
            # compaction (k_vox_compact: 15 -> 187 us): it runs first, alone, and the pool work is forked behind it, next to
            # the insert / window launches that are small and separated by read-backs
            # Its launches go out first; the pool's launches (a side stream that waits for them) are prepared on the host
            # while they execute, and only then the down-sampling's round trip is made.
            from .tools import voxel_down_sample_async, voxel_down_sample_finish, voxel_down_sample_launch, voxel_down_sample_torch

            two_phase = update_points.is_cuda and update_points.shape[0] > 0
            cmp_dev = self._cmp_counts if defer_cmp else None  # [kept rows, near-surface rows] of the compaction, on the device
            # The map growth's voxel pass stays in flight where the insert + window that follow take its list and count on the
            # device: its count, the compaction's two and the insert / window counts then come back in ONE read-back
            # (NeuralPoints.update); the two-phase path (a round trip of its own for the voxel count) otherwise.
            main = torch.cuda.current_stream(coord.device)
            side = getattr(self, "_side_stream", None)
            if side is None or side.device != coord.device:
                side = self._side_stream = _lib.low_priority_stream(coord.device)

            def fork_pool():
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    self._pool_append_filter_fused(coord, gcoord, sdf_label, weight, stamp, cur_pose_torch, frame_id, defer=True,
                                                   n_b_dev=None if cmp_dev is None else cmp_dev[0:1])

            pool_first = two_phase and defer_cmp and async_upd and os.environ.get("CLID_POOL_FORK_EARLY", "1") != "0"
            # CLID_POOL_GATE=1: the pool's flag / list / drop passes beside the voxel pass, its compaction (which takes every wave
            # slot: k_vox_bucket_sort 15 -> 120 us beside it) only behind the voxel pass, beside the insert / window launches
            gate = pool_first and os.environ.get("CLID_POOL_GATE", POOL_GATE_DEFAULT) == "1"
            if pool_first and not gate:
                # Nothing waits for the voxel pass on the host, and the frame's critical path runs through the pool (flags, list,
                # capacity drop, count, 220 us of compaction = 330 us, then the new-sample selection on its output): it is
                # forked right behind the sampler's compaction, its bandwidth-bound prelude next to the voxel pass
                fork_pool()
            if two_phase and defer_cmp and async_upd:
                if gate:
                    fork_at = main.record_event()  # (the side stream must not wait for the voxel pass: fork where the pool's inputs are ready)
                vox_idx = voxel_down_sample_async(update_points, nm.resolution, nm.update_counts(coord.device)[3:5], n_dev=cmp_dev[1:2])
                if gate:
                    vox_done = torch.cuda.Event()
                    vox_done.record(main)
                    side.wait_event(fork_at)
                    with torch.cuda.stream(side):
                        self._pool_append_filter_fused(coord, gcoord, sdf_label, weight, stamp, cur_pose_torch, frame_id, defer=True,
                                                       n_b_dev=None if cmp_dev is None else cmp_dev[0:1], scatter_after=vox_done)
            pending_vox = (voxel_down_sample_launch(update_points, nm.resolution, n_dev=None if cmp_dev is None else cmp_dev[1:2])
                           if two_phase and vox_idx is None else None)
            if not pool_first:
                fork_pool()
            if vox_idx is not None:
                keep_idx = None
            else:
                keep_idx = voxel_down_sample_finish(pending_vox) if two_phase else voxel_down_sample_torch(update_points, nm.resolution)
            if defer_cmp and vox_idx is None:  # (the stream has just been drained by the voxel count's round trip: this one finds its data ready)
                kept, n_near = _lib.read_counts(self._cmp_counts, 2)
                coord, gcoord, sdf_label, weight, stamp = coord[:kept], gcoord[:kept], sdf_label[:kept], weight[:kept], stamp[:kept]
                update_points = update_points[:n_near]
                n_cur = self.cur_sample_count = kept
                defer_cmp = False
            nm._presampled = (update_points, (vox_idx, nm.update_counts(coord.device)[3:5], None if cmp_dev is None else cmp_dev[1:2])
                              if vox_idx is not None else update_points[keep_idx])
        if defer_cmp and vox_idx is None:  # (not reached with the conditions above; kept for safety: settle the counts before anything uses the rows)
            kept, n_near = _lib.read_counts(self._cmp_counts, 2)
            coord, gcoord, sdf_label, weight, stamp = coord[:kept], gcoord[:kept], sdf_label[:kept], weight[:kept], stamp[:kept]
            update_points = update_points[:n_near]
            n_cur = self.cur_sample_count = kept
            defer_cmp = False
        prefetched = [False]
        if (overlap and os.environ.get("CLID_TABLE_PREFETCH", "1") != "0" and os.environ.get("CLID_TABLE_PREFETCH_EARLY", "1") != "0"
                and os.environ.get("CLID_TABLE_PREFETCH_HOOK", "1") != "0"):
            # the table + cell directory of the mapping() call that follows depend on the new window only: their build goes out
            # from INSIDE NeuralPoints.update, right behind the window's count read-back (the device idles there until the host
            # has something for it), on the third stream ordered behind the window's launches
            third = getattr(self, "_third_stream", None)
            if third is None or third.device != coord.device:
                third = self._third_stream = torch.cuda.Stream(device=coord.device)
            main_s = torch.cuda.current_stream(coord.device)

            def _prefetch_now():
                third.wait_event(main_s.record_event())
                nm.prefetch_local_table(third)
                prefetched[0] = True

            nm._on_window_ready = _prefetch_now
        self.cur_new_point_ratio = nm.update(update_points, origin, orientation, frame_id)
        nm.__dict__.pop("_on_window_ready", None)  # (a path that never reached the fused window selection)
        if defer_cmp:  # the compaction's counts came back with the insert / window counts
            kept, n_near = nm._last_update_counts[5:7]
            coord, gcoord, sdf_label, weight, stamp = coord[:kept], gcoord[:kept], sdf_label[:kept], weight[:kept], stamp[:kept]
            update_points = update_points[:n_near]
            n_cur = self.cur_sample_count = kept
            defer_cmp = False

        new_pending = None
        if overlap:
            # Tail of the frame, in the order that keeps the device busy: the local probe table + cell directory of the mapping()
            # call that follows (csrc/table.hip, csrc/celldir.hip: ~100 us of small launches at 250 k local points) depend on
            # the map only -- enqueued first, on a third stream that waits for the map update alone (not, through `main`, for
            # the pool's compaction on the side stream); the new-sample selection, which needs the pool arrays, next to it on
            # `main`; the host-only bookkeeping while both run; then the frame's last read-back.
            if os.environ.get("CLID_TABLE_PREFETCH", "1") != "0":
                third = getattr(self, "_third_stream", None)
                if third is None or third.device != coord.device:
                    third = self._third_stream = torch.cuda.Stream(device=coord.device)
                if prefetched[0]:  # (went out from inside NeuralPoints.update)
                    main.wait_stream(side)
                elif os.environ.get("CLID_TABLE_PREFETCH_EARLY", "1") != "0":
                    third.wait_event(main.record_event())
                    nm.prefetch_local_table(third)
                    main.wait_stream(side)
                else:
                    main.wait_stream(side)
                    third.wait_stream(main)
                    nm.prefetch_local_table(third)
            else:
                main.wait_stream(side)
            if cfg.bs_new_sample > 0:
                # the new-sample selection (below) launched on the pool arrays still in flight: it reads the two pool counts on
                # the device, so ONE read-back serves the pool maintenance and the selection
                new_pending = self._new_sample_launch_pending(coord.shape[0])
            self.determine_used_pose()
            self._pool_filter_finish(with_tail=new_pending is not None)
        elif fused_pool:
            self.determine_used_pose()
            self._pool_append_filter_fused(coord, gcoord if gcoord is not None else transform_torch(coord, cur_pose_torch),
                                           sdf_label, weight, stamp, cur_pose_torch, frame_id)
        else:
            self.determine_used_pose()
            self._pool_append_filter_torch(coord, sdf_label, weight, stamp, sem_label, color_label, normal_label,
                                           cur_pose_torch, origin, frame_id, n_cur)

        if getattr(self.local_point_cloud_map, "_count_pending", False):  # (paths without the merged tail read-back)
            self.local_point_cloud_map._finish_count()
        # newly observed region: samples of this frame whose neighbourhood is still uncertain (:400-462)
        if cfg.bs_new_sample > 0:
            if new_pending is not None:
                self.new_idx = new_pending[: self._new_count_host]  # (launched above, read with the pool counts)
            else:
                cur = self.global_coord_pool[self.global_coord_pool.shape[0] - self.cur_sample_count:]
                cur_label = self.sdf_label_pool[self.sdf_label_pool.shape[0] - self.cur_sample_count:]
                nm.set_search_neighborhood(num_nei_cells=1, search_alpha=0.0)
                try:
                    if self._new_sample_fused_ok(cur, cur_label):
                        self.new_idx = self._new_sample_select_fused(cur, cur_label)
                    else:
                        certainty = torch.zeros(cur.shape[0], device=cur.device)
                        for head in range(0, cur.shape[0], cfg.infer_bs):
                            certainty[head:head + cfg.infer_bs] = nm.query_certainty(cur[head:head + cfg.infer_bs, :])
                        self.new_idx = torch.where(
                            (certainty < getattr(cfg, "new_certainty_thre", 1.0)) & (torch.abs(cur_label) < cfg.surface_sample_range_m * 3.0)
                        )[0]
                        self.new_idx += self.pool_sample_count - self.cur_sample_count
                finally:
                    nm.set_search_neighborhood(num_nei_cells=cfg.num_nei_cells, search_alpha=cfg.search_alpha)
            self.adaptive_iter_offset = 0
            ratio = self.new_idx.shape[0] / max(self.cur_sample_count, 1)
            if cfg.adaptive_iters:
                if ratio < getattr(cfg, "new_sample_ratio_less", 0.02):
                    self.adaptive_iter_offset = -5
                elif ratio > getattr(cfg, "new_sample_ratio_more", 0.15):
                    self.adaptive_iter_offset = 5
                    if frame_id > cfg.freeze_after_frame and ratio > getattr(cfg, "new_sample_ratio_restart", 0.3):
                        self.adaptive_iter_offset = 10

    def _sample_compact_fused(self, pts, cur_pose_torch, frame_id, defer_counts=False, counts=None):
        """Sampler launch + `clid_sample_compact`: (coord, gcoord, sdf_label, weight, stamp, update_points) of this frame
        as utils/mapper.py:240-283 / :297-310 produce them (kept rows in order; update_points = world-frame rows with
        |sdf| < surface_sample_range_m * map_surface_ratio)."""
        cfg = self.config
        lib = _lib.load()
        coord, label, weight, keep, _ = self.sampler._run(pts, self.local_point_cloud_map, cur_pose_torch, None)
        n, dev = coord.shape[0], coord.device
        need = int(lib.clid_sample_compact_workspace_bytes(n))
        if getattr(self, "_cmp_ws", None) is None or self._cmp_ws.numel() < need or self._cmp_ws.device != dev:
            self._cmp_ws = torch.empty(int(need * 1.3) + 256, device=dev, dtype=torch.uint8)
            self._cmp_own_counts = torch.zeros(2, device=dev, dtype=torch.int64)
        # [kept rows | near-surface rows]: in the caller's block when it reads them with other counts in one read-back
        self._cmp_counts = counts if counts is not None else self._cmp_own_counts
        out3 = torch.empty((3, max(n, 1), 3), device=dev, dtype=torch.float32)   # coord | gcoord | update_points
        out1 = torch.empty((2, max(n, 1)), device=dev, dtype=torch.float32)      # label | weight
        stamp = torch.empty(max(n, 1), device=dev, dtype=torch.int32)
        T = cur_pose_torch.detach().to("cpu", torch.float32)  # transform_torch casts the pose to the points' dtype
        pose = (C.c_float * 12)(*T[:3, :].reshape(-1).tolist())
        near = float(cfg.surface_sample_range_m * getattr(cfg, "map_surface_ratio", 0.5))
        _lib.check(lib.clid_sample_compact(coord.data_ptr(), label.data_ptr(), weight.data_ptr(), keep.data_ptr(), n, pose, near,
                                           int(frame_id), out3[0].data_ptr(), out3[1].data_ptr(), out1[0].data_ptr(),
                                           out1[1].data_ptr(), stamp.data_ptr(), out3[2].data_ptr(),
                                           self._cmp_counts.data_ptr(), self._cmp_ws.data_ptr(), _lib.stream()),
                   "clid_sample_compact")
        if defer_counts:
            # the caller reads the two counts later (with the map growth's voxel count) and slices then: full-capacity tensors,
            # the device counts for the launches in between
            return out3[0], out3[1], out1[0], out1[1], stamp, out3[2]
        kept, n_near = _lib.read_counts(self._cmp_counts, 2)  # the one host round trip (sizes the views)
        return out3[0][:kept], out3[1][:kept], out1[0][:kept], out1[1][:kept], stamp[:kept], out3[2][:n_near]

    def _new_sample_fused_ok(self, cur, cur_label):
        nm = self.neural_points
        big = nm.buffer_pt_index
        return (cur.is_cuda and cur.dtype == torch.float32 and cur_label.dtype == torch.float32 and big is not None and big.is_cuda
                and big.dtype == torch.int64 and int(nm.buffer_size) < (1 << 30) and nm.point_certainties.dtype == torch.float32
                and nm.neural_points.dtype == torch.float32 and os.environ.get("CLID_FUSED_NEWSEL", "1") != "0")

    def _new_sample_launch_pending(self, n_upper: int):
        """`clid_new_sample_select` on the output arrays of the pool call that is still in flight (the frame's samples are
        their last counts[1] rows, read on the device); returns the index buffer, or None when the fused path does not apply.
        The count is read after the pool's (`process_frame`)."""
        cfg, nm = self.config, self.neural_points
        out = self._pool_pending[1]
        if not self._new_sample_fused_ok(out["gcoord"], out["label"]) or n_upper <= 0:
            return None
        lib = _lib.load()
        dev = out["gcoord"].device
        need = int(lib.clid_new_sample_workspace_bytes(n_upper))
        if getattr(self, "_new_ws", None) is None or self._new_ws.numel() < need or self._new_ws.device != dev:
            self._new_ws = torch.empty(int(need * 1.3) + 256, device=dev, dtype=torch.uint8)
        self._frame_count_block(dev)
        idx = torch.empty(n_upper, device=dev, dtype=torch.int64)
        nm.set_search_neighborhood(num_nei_cells=1, search_alpha=0.0)
        try:
            if nm._delta.device != dev:
                nm._delta = nm._delta.to(dev)
            _lib.check(lib.clid_new_sample_select(
                nm.buffer_pt_index.data_ptr(), int(nm.buffer_size), _lib.require_cuda(nm.neural_points, "neural_points", torch.float32).data_ptr(),
                nm.point_certainties.contiguous().data_ptr(), nm._delta.data_ptr(), int(nm.neighbor_K), float(nm.resolution),
                float(nm.max_valid_dist2), out["gcoord"].data_ptr(), out["label"].data_ptr(), n_upper,
                float(getattr(cfg, "new_certainty_thre", 1.0)), float(cfg.surface_sample_range_m * 3.0), 0, idx.data_ptr(),
                self._new_count.data_ptr(), self._pool_counts.data_ptr(), self._new_ws.data_ptr(), _lib.stream()),
                "clid_new_sample_select")
        finally:
            nm.set_search_neighborhood(num_nei_cells=cfg.num_nei_cells, search_alpha=cfg.search_alpha)
        return idx

    def _new_sample_select_fused(self, cur, cur_label):
        """utils/mapper.py:409-423 in one enqueue (`clid_new_sample_select`): certainty probe of the global map with the
        current stencil, the two tests, the ascending pool indices of the selected samples; ONE read-back (their count)."""
        cfg, nm = self.config, self.neural_points
        lib = _lib.load()
        n, dev = int(cur.shape[0]), cur.device
        need = int(lib.clid_new_sample_workspace_bytes(n))
        if getattr(self, "_new_ws", None) is None or self._new_ws.numel() < need or self._new_ws.device != dev:
            self._new_ws = torch.empty(int(need * 1.3) + 256, device=dev, dtype=torch.uint8)
        self._frame_count_block(dev)
        idx = torch.empty(max(n, 1), device=dev, dtype=torch.int64)
        if nm._delta.device != dev:
            nm._delta = nm._delta.to(dev)
        x, lab = cur.contiguous(), cur_label.contiguous()
        _lib.check(lib.clid_new_sample_select(
            nm.buffer_pt_index.data_ptr(), int(nm.buffer_size), _lib.require_cuda(nm.neural_points, "neural_points", torch.float32).data_ptr(),
            nm.point_certainties.contiguous().data_ptr(), nm._delta.data_ptr(), int(nm.neighbor_K), float(nm.resolution),
            float(nm.max_valid_dist2), x.data_ptr(), lab.data_ptr(), n, float(getattr(cfg, "new_certainty_thre", 1.0)),
            float(cfg.surface_sample_range_m * 3.0), int(self.pool_sample_count - self.cur_sample_count), idx.data_ptr(),
            self._new_count.data_ptr(), None, self._new_ws.data_ptr(), _lib.stream()), "clid_new_sample_select")
        return idx[: _lib.read_counts(self._new_count, 1)[0]]

    def _pool_append_filter_torch(self, coord, sdf_label, weight, stamp, sem_label, color_label, normal_label,
                                  cur_pose_torch, origin, frame_id, n_cur):
        """utils/mapper.py:297-392 as torch ops (configs with semantic / colour / normal pools, a pending pose-graph
        re-projection, pool_filter_freq > 1, or CPU tensors)."""
        from .tools import transform_torch

        cfg = self.config
        # pool append (:297-333)
        self.coord_pool = torch.cat((self.coord_pool, coord), 0)
        self.weight_pool = torch.cat((self.weight_pool, weight), 0)
        self.sdf_label_pool = torch.cat((self.sdf_label_pool, sdf_label), 0)
        self.time_pool = torch.cat((self.time_pool, stamp), 0)
        self.sem_label_pool = None if sem_label is None else (
            sem_label if self.sem_label_pool is None else torch.cat((self.sem_label_pool, sem_label), 0))
        self.color_pool = None if color_label is None else (
            color_label if self.color_pool is None else torch.cat((self.color_pool, color_label), 0))
        self.normal_label_pool = None if normal_label is None else (
            normal_label if self.normal_label_pool is None else torch.cat((self.normal_label_pool, normal_label), 0))
        if self.ba_done_flag:  # poses of old frames moved: re-project the whole pool (:322-327)
            from .tools import transform_batch_torch

            self.global_coord_pool = transform_batch_torch(self.coord_pool, self.used_poses[self.time_pool.long()])
            self.ba_done_flag = False
        else:
            self.global_coord_pool = torch.cat((self.global_coord_pool, transform_torch(coord, cur_pose_torch)), 0)

        # window / capacity filter (:337-392)
        if (frame_id + 1) % getattr(cfg, "pool_filter_freq", 1) == 0:
            rel = self.global_coord_pool - origin  # float64 when the pose is (type promotion, as in the reference)
            keep = torch.sum(rel**2, dim=-1) < cfg.window_radius**2
            kept = torch.nonzero(keep).squeeze(1)
            if kept.shape[0] > cfg.pool_capacity:
                drop = torch.randint(0, kept.shape[0], (kept.shape[0] - cfg.pool_capacity,), device=keep.device,
                                     generator=_lib.replica_generator(self, cfg, keep.device, 3))
                keep[kept[drop]] = False
                kept = torch.nonzero(keep).squeeze(1)
            total = keep.shape[0]  # `kept` (one host round trip) sizes all pools
            self.coord_pool = self.coord_pool.index_select(0, kept)
            self.global_coord_pool = self.global_coord_pool.index_select(0, kept)
            self.sdf_label_pool = self.sdf_label_pool.index_select(0, kept)
            self.weight_pool = self.weight_pool.index_select(0, kept)
            self.time_pool = self.time_pool.index_select(0, kept)
            if self.normal_label_pool is not None:
                self.normal_label_pool = self.normal_label_pool.index_select(0, kept)
            if self.sem_label_pool is not None:
                self.sem_label_pool = self.sem_label_pool.index_select(0, kept)
            if self.color_pool is not None:
                self.color_pool = self.color_pool.index_select(0, kept)
            self.cur_sample_count = int((kept >= total - n_cur).sum().item()) if n_cur > 0 else 0
            self.pool_sample_count = kept.shape[0]
        else:
            self.cur_sample_count = n_cur
            self.pool_sample_count = self.coord_pool.shape[0]


    def _pool_append_filter_fused(self, coord, gcoord, sdf_label, weight, stamp, cur_pose_torch, frame_id, defer=False,
                                  n_b_dev=None, scatter_after=None):
        """utils/mapper.py:297-392 in one enqueue (csrc/mapops.hip clid_pool_filter): append this frame's samples, window
        test in float64, random drop above `pool_capacity`, stable compaction of the five arrays into the other half of a
        ping-pong buffer; ONE small read-back for the two counts the host needs (pool size for the batch draws, number
        of this frame's samples that stayed)."""
        lib = _lib.load()
        cfg = self.config
        dev = coord.device
        n_a, n_b = int(self.coord_pool.shape[0]), int(coord.shape[0])
        n = n_a + n_b
        bufs = getattr(self, "_pool_bufs", None)
        if bufs is None or any(bf is not None and bf["coord"].device != dev for bf in bufs):
            bufs = self._pool_bufs = [None, None]
            self._pool_side = 0
        side = 1 - self._pool_side
        cap_rows = max(n, min(int(cfg.pool_capacity) + 2 * n_b, 2 * n) + 1024)
        if bufs[side] is None or bufs[side]["coord"].shape[0] < n:
            bufs[side] = {"coord": torch.empty((cap_rows, 3), device=dev), "gcoord": torch.empty((cap_rows, 3), device=dev),
                          "label": torch.empty(cap_rows, device=dev), "weight": torch.empty(cap_rows, device=dev),
                          "time": torch.empty(cap_rows, device=dev, dtype=torch.int32)}
        out = bufs[side]
        need = int(lib.clid_pool_workspace_bytes(n))
        if getattr(self, "_pool_ws", None) is None or self._pool_ws.numel() < need or self._pool_ws.device != dev:
            self._pool_ws = torch.empty(int(need * 1.2) + 256, device=dev, dtype=torch.uint8)
        self._frame_count_block(dev)
        a = [_lib.require_cuda(t, nme, dt) for t, nme, dt in (
            (self.coord_pool, "coord_pool", torch.float32), (self.global_coord_pool, "global_coord_pool", torch.float32),
            (self.sdf_label_pool, "sdf_label_pool", torch.float32), (self.weight_pool, "weight_pool", torch.float32),
            (self.time_pool, "time_pool", torch.int32))]
        b = [t.contiguous() for t in (coord.float(), gcoord.float(), sdf_label.float(), weight.float(), stamp.to(torch.int32))]
        origin = (C.c_double * 3)(*[float(v) for v in cur_pose_torch[:3, 3].tolist()])  # the pose is already on the host
        self._pool_drop_seed = (getattr(self, "_pool_drop_seed", int(getattr(cfg, "seed", 42)) * 7919) * 6364136223846793005 + 1442695040888963407) % (1 << 64)
        # scatter_after: a recorded torch.cuda.Event the compaction launch waits for (the passes in front of it do not)
        _lib.check(lib.clid_pool_filter_after(
            a[0].data_ptr(), a[1].data_ptr(), a[2].data_ptr(), a[3].data_ptr(), a[4].data_ptr(), n_a,
            b[0].data_ptr(), b[1].data_ptr(), b[2].data_ptr(), b[3].data_ptr(), b[4].data_ptr(), n_b,
            origin, float(cfg.window_radius) ** 2, int(cfg.pool_capacity), self._pool_drop_seed,
            out["coord"].data_ptr(), out["gcoord"].data_ptr(), out["label"].data_ptr(), out["weight"].data_ptr(),
            out["time"].data_ptr(), self._pool_counts.data_ptr(), self._pool_ws.data_ptr(), _lib.ptr(n_b_dev), _lib.stream(),
            None if scatter_after is None else scatter_after.cuda_event), "clid_pool_filter_after")
        self._pool_pending = (side, out, (a, b), scatter_after)  # (inputs stay referenced until the launches have been joined)
        if not defer:
            self._pool_filter_finish()

    def _frame_count_block(self, dev):
        """One device block for the data-dependent counts of a frame's tail -- [pool kept, kept of this frame, -, new samples,
        raw-point map size, -, -, -] -- so that ONE read-back serves the pool maintenance, the new-sample selection and the
        raw-point map (whose count nothing needs before the end of process_frame)."""
        fc = getattr(self, "_frame_counts", None)
        if fc is None or fc.device != torch.device(dev):
            fc = self._frame_counts = torch.zeros(16, device=dev, dtype=torch.int64)  # [8:10]: the scan's voxel pass [count | failed]
            self._pool_counts, self._new_count = fc[0:3], fc[3:4]
        return fc

    def _pool_filter_finish(self, with_tail: bool = False):
        side, out = self._pool_pending[:2]
        self._pool_pending = None
        if with_tail:  # the new-sample selection was launched on the pool arrays in flight; the raw-point map's count rides along
            got = _lib.read_counts(self._frame_counts, 10)
            kept, kept_cur = got[0], got[1]
            self._new_count_host = got[3]
            lpm = self.local_point_cloud_map
            if getattr(lpm, "_count_pending", False):
                lpm._finish_count(got[4], got[9])
        else:
            kept, kept_cur = _lib.read_counts(self._pool_counts, 2)  # the one host round trip of the pool maintenance
        self._pool_side = side
        self.coord_pool, self.global_coord_pool = out["coord"][:kept], out["gcoord"][:kept]
        self.sdf_label_pool, self.weight_pool, self.time_pool = out["label"][:kept], out["weight"][:kept], out["time"][:kept]
        self.cur_sample_count = int(kept_cur)
        self.pool_sample_count = int(kept)

    def bundle_adjustment(self, iter_count, window_size: int = 50, use_lie_group: bool = False):
        """utils/mapper.py:866-965 (needs pypose; ba_freq_frame = 0 in every shipped config)."""
        raise NotImplementedError("bundle_adjustment is disabled in all shipped configs (ba_freq_frame=0) and out of scope")

    def get_data_pool_o3d(self, down_rate=1, only_cur_data=False):
        """utils/mapper.py:553-600: the training pool as a point cloud for the GUI, coloured by the SDF label (red
        behind / blue in front of the surface).  open3d PointCloud when installed, else tools.PointCloudArrays."""
        import numpy as np

        from .tools import point_cloud_o3d

        if only_cur_data:
            sl = slice(self.global_coord_pool.shape[0] - self.cur_sample_count, None, 3)
        else:
            sl = slice(None, None, max(int(down_rate), 1))
        pts = self.global_coord_pool[sl].detach().cpu().numpy().astype(np.float64)
        colors = None
        if self.sdf_label_pool is not None:
            lab = self.sdf_label_pool[sl].detach().cpu().numpy().astype(np.float64)
            t = np.clip(np.abs(lab) / max(float(self.config.free_sample_end_dist_m), 1e-6), 0.0, 1.0)
            colors = np.zeros((lab.shape[0], 3))
            colors[lab < 0, 0] = 1.0 - 0.5 * t[lab < 0]   # behind the surface: red
            colors[lab >= 0, 2] = 1.0 - 0.5 * t[lab >= 0]  # in front: blue
        return point_cloud_o3d(pts, colors)
