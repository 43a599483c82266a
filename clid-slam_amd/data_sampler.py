"""Per-frame training-sample generation ("next" row N2).

Drop-in for `DataSampler` (utils/data_sampler.py:11-402): `sample` (region-specific SDF labels, CLID-SLAM's
own sampler) and `sample_pin` (projective labels) return what the reference methods return.  The reference
composes each of them from ~60 elementwise torch ops over [rays x 8] tensors; here one HIP launch
(`k_sample_frame`, csrc/sampler.hip) writes every sample directly in the ray-major output order and runs the
region-specific estimate in the same thread.  The random draws come from torch's device generator in the
reference's own order (randn for the near-surface offsets, rand for the front / behind free-space samples),
so a seeded run consumes the generator exactly as the reference does on the same device; `noise=` injects
the draws instead (tests).
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib


class DataSampler:
    def __init__(self, config):
        self.config = config
        self.dev = config.device

    # ------------------------------------------------------------------
    def _run(self, points_torch, cloud, pose, noise):
        cfg = self.config
        lib = _lib.load()
        pts = _lib.require_cuda(points_torch.detach().to(torch.float32).contiguous(), "points_torch", torch.float32)
        dev = pts.device
        R = pts.shape[0]
        ns, nf, nb = int(cfg.surface_sample_n), int(cfg.free_front_n), int(cfg.free_behind_n)
        n_all = 1 + ns + nf + nb
        pre = getattr(self, "_predrawn", None)
        self._predrawn = None
        if noise is None and pre is not None and pre[0] == (R, str(dev)):
            z_s, u_f, u_b = pre[1]
        elif noise is None:  # utils/data_sampler.py:47, :72, :93 (this order)
            if pre is not None:
                # a predraw for another ray count / device: the generator goes back to where it was before it, so the
                # stream of random numbers stays the reference's (and the same on every rank of a data-parallel group)
                self._restore_generator(pre[2])
            z_s, u_f, u_b = self._draw(R, dev)
        else:
            z_s, u_f, u_b = (t.to(dev, torch.float32).contiguous() for t in noise)
        p = _lib.SamplerParams()
        p.surface_sample_range_m = float(cfg.surface_sample_range_m)
        p.free_sample_begin_ratio = float(cfg.free_sample_begin_ratio)
        p.free_sample_end_dist_m = float(cfg.free_sample_end_dist_m)
        p.dist_weight_scale, p.max_range = float(cfg.dist_weight_scale), float(cfg.max_range)
        p.surface_sample_n, p.free_front_n, p.free_behind_n = ns, nf, nb
        p.dist_weight_on, p.behind_dropoff_on = int(bool(cfg.dist_weight_on)), int(bool(cfg.behind_dropoff_on))
        T = torch.eye(4) if pose is None else torch.as_tensor(pose).detach().cpu()
        T = T.to(torch.float32)  # transform_torch casts the pose to the points' dtype (utils/tools.py:604)
        p.pose = (C.c_float * 12)(*T[:3, :].reshape(-1).tolist())
        coord = torch.empty((R * n_all, 3), device=dev, dtype=torch.float32)
        label = torch.empty(R * n_all, device=dev, dtype=torch.float32)
        weight = torch.empty(R * n_all, device=dev, dtype=torch.float32)
        keep = torch.empty(R * n_all, device=dev, dtype=torch.uint8)
        view, alive = (None, None) if cloud is None else cloud._cloud_view()
        _lib.check(
            lib.clid_sample_frame(None if view is None else C.byref(view), C.byref(p), pts.data_ptr(), R,
                                  _lib.ptr(z_s), _lib.ptr(u_f), _lib.ptr(u_b), coord.data_ptr(), label.data_ptr(),
                                  weight.data_ptr(), keep.data_ptr(), _lib.stream()),
            "clid_sample_frame",
        )
        return coord, label, weight, keep, n_all

    def _draw(self, R, dev):
        cfg = self.config
        ns, nf, nb = int(cfg.surface_sample_n), int(cfg.free_front_n), int(cfg.free_behind_n)
        gen = _lib.replica_generator(self, cfg, dev, 1)  # None (global RNG, as the reference) unless data-parallel
        z_s = torch.randn(R * ns, 1, device=dev, generator=gen)
        u_f = torch.rand(R * nf, 1, device=dev, generator=gen)
        u_b = torch.rand(R * nb, 1, device=dev, generator=gen)
        return z_s, u_f, u_b

    def predraw(self, n_rays: int, dev) -> None:
        """The next `_run`'s random draws for `n_rays` rays, enqueued ahead of time (Mapper.process_frame: before the raw-point
        map update, whose round trips they then overlap); consumed by the next `_run` with that ray count, dropped otherwise."""
        dev = torch.device(dev) if not isinstance(dev, torch.device) else dev
        gen = _lib.replica_generator(self, self.config, dev, 1)
        state = (gen, gen.get_state()) if gen is not None else (None, torch.cuda.get_rng_state(dev) if dev.type == "cuda" else torch.get_rng_state())
        self._predrawn = ((int(n_rays), str(dev)), self._draw(int(n_rays), dev), (dev, state))

    @staticmethod
    def _restore_generator(saved) -> None:
        dev, (gen, state) = saved
        if gen is not None:
            gen.set_state(state)
        elif dev.type == "cuda":
            torch.cuda.set_rng_state(state, dev)
        else:
            torch.set_rng_state(state)

    def sample(self, points_torch, local_point_cloud_map, cur_pose_torch, noise=None):
        """utils/data_sampler.py:260-402: (coord [S,3] sensor frame, sdf_label [S], weight [S]); near-surface
        samples carry sign x region-specific |SDF| and are dropped where no raw map point is around them."""
        coord, label, weight, keep, _ = self._run(points_torch, local_point_cloud_map, cur_pose_torch, noise)
        kept = torch.nonzero(keep).flatten()  # one host round trip for the three outputs
        return coord.index_select(0, kept), label.index_select(0, kept), weight.index_select(0, kept)

    def sample_pin(self, points_torch, normal_torch=None, sem_label_torch=None, color_torch=None, noise=None):
        """utils/data_sampler.py:16-258: 6-tuple (coord, sdf_label, normal_label, sem_label, color_label, weight)
        with projective labels.  Normal / semantic / colour labels are plain repeats of per-ray inputs."""
        coord, label, weight, _, n_all = self._run(points_torch, None, None, noise)
        R = points_torch.shape[0]
        ns, nf, nb = int(self.config.surface_sample_n), int(self.config.free_front_n), int(self.config.free_behind_n)
        normal = None if normal_torch is None else normal_torch.repeat_interleave(n_all, dim=0)
        sem = None
        if sem_label_torch is not None:  # free-space samples carry label 0 (:181-192)
            sem = torch.zeros((R, n_all), dtype=torch.int, device=coord.device)
            sem[:, : 1 + ns] = sem_label_torch.reshape(R, 1).int()
            sem = sem.reshape(-1)
        color = None
        if color_torch is not None:  # ... and zero colour (:195-205)
            ch = color_torch.shape[1]
            color = torch.zeros((R, n_all, ch), dtype=color_torch.dtype, device=coord.device)
            color[:, : 1 + ns, :] = color_torch.reshape(R, 1, ch)
            color = color.reshape(-1, ch)
        return coord, label, normal, sem, color, weight
