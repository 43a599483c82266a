"""Tracking measurement model ("next" row N1): the consumer of the analytic
d SDF / d x inference path, `IEKFOM.h_model` (utils/error_state_iekf.py:176-264), as ONE fused HIP launch.

    z, H, valid_points, R_inv = h_model(neural_points, geo_decoder, config, rot, pos, pc_imu)
    S, HtRz, n_valid = normal_equations(neural_points, geo_decoder, config, rot, pos, pc_imu)

`h_model` returns what the reference method returns (plus `R_inv`, which the reference stores on `self`).
`normal_equations` returns the only quantities `update_iterated` (:299-305) derives from H:
S = H^T R_inv H (18 x 18, float64, non-zero 6 x 6 block) and H^T R_inv z, reduced on the GPU without
materialising H.  The 18-state filter itself (predict / boxplus / covariance) is host-side 18 x 18 math and
stays with the reference (out of scope).
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib


def _launch(neural_points, geo_decoder, config, rot, pos, pc_imu, per_point: bool, reduce: bool):
    lib = _lib.load()
    x = _lib.require_cuda(pc_imu.detach().to(torch.float32).contiguous(), "pc_imu", torch.float32)
    n = x.shape[0]
    dev = x.device
    view, keep = neural_points._map_view(True)
    W1, b1, W2, b2 = geo_decoder.flat_params()
    # the state lives on the device in the reference's filter (utils/error_state_iekf.py:176-186): the kernel reads the 12
    # numbers there (clid_track_model_dev) -- two tiny conversions, no host round trip per evaluation; a host pose goes by value
    rot, pos = torch.as_tensor(rot), torch.as_tensor(pos)
    on_dev = rot.is_cuda and pos.is_cuda
    if on_dev:
        r = rot.detach().to(torch.float32).contiguous()
        t = pos.detach().to(torch.float32).contiguous()
    else:
        r = (C.c_float * 9)(*rot.detach().to(torch.float32).reshape(-1).tolist())
        t = (C.c_float * 3)(*pos.detach().to(torch.float32).reshape(-1).tolist())
    out = {}
    if per_point:
        # raw per-point outputs: a cached scratch set per (n, device) -- everything handed to the caller below is derived
        # by masked indexing, i.e. copied (the filter calls this 5-20 times per scan)
        cache = neural_points.__dict__.setdefault("_track_scratch", {})
        out = cache.get((n, str(dev)))
        if out is None:
            if len(cache) >= 4:
                cache.clear()
            out = cache[(n, str(dev))] = {
                "sdf": torch.empty(n, device=dev, dtype=torch.float32), "grad": torch.empty((n, 3), device=dev, dtype=torch.float32),
                "pmap": torch.empty((n, 3), device=dev, dtype=torch.float32), "valid": torch.empty(n, device=dev, dtype=torch.int32)}
    ne = None
    if reduce:  # CLID_TRACK_COPIES partial copies of the 28 sums; two buffers alternate so a caller may still hold the last result
        ring = neural_points.__dict__.setdefault("_track_ne", {})
        pair = ring.get(str(dev))
        if pair is None:
            pair = ring[str(dev)] = [torch.empty((16, 32), device=dev, dtype=torch.float64) for _ in range(2)] + [0]
        pair[2] ^= 1
        ne = pair[pair[2]].zero_()
    fn = lib.clid_track_model_dev if on_dev else lib.clid_track_model
    _lib.check(
        fn(C.byref(view), _lib.ptr(W1), _lib.ptr(b1), _lib.ptr(W2), _lib.ptr(b2),
                             float(geo_decoder.sdf_scale), _lib.ptr(r) if on_dev else r, _lib.ptr(t) if on_dev else t, int(config.track_mask_query_nn_k),
                             float(config.reg_min_grad_norm), float(config.reg_max_grad_norm),
                             float(config.surface_sample_range_m * getattr(config, "max_sdf_std_ratio", 1.0)), _lib.ptr(x), n,
                             _lib.ptr(out.get("sdf")), _lib.ptr(out.get("grad")), _lib.ptr(out.get("pmap")),
                             _lib.ptr(out.get("valid")), _lib.ptr(ne), _lib.stream()),
        "clid_track_model",
    )
    return x, out, ne


def h_model(neural_points, geo_decoder, config, rot, pos, pc_imu):
    """(sdf_residual [Nv] f64, H [Nv,18] f64, valid_points [Nv,3], R_inv [Nv] f64) as
    utils/error_state_iekf.py:176-264 computes them for the state (rot, pos)."""
    x, out, _ = _launch(neural_points, geo_decoder, config, rot, pos, pc_imu, True, False)
    tdt = getattr(config, "tran_dtype", torch.float64)
    valid = out["valid"].bool()
    g = out["grad"][valid]
    p = x[valid]
    rot32 = torch.as_tensor(rot).to(x.device, torch.float32)
    q = g @ rot32  # rows: R^T g
    H = torch.zeros((g.shape[0], 18), device=x.device, dtype=tdt)
    H[:, 0:3] = torch.cross(p, q, dim=1)  # -g^T R [p]x
    H[:, 3:6] = g
    z = out["sdf"][valid].to(tdt)
    ga = (g.norm(dim=-1) - 1.0).to(tdt)
    r_inv = 1 / (1 + ga**2) * (0.4 / (0.4 + z**2)) * 1000
    return z, H, out["pmap"][valid], r_inv


def normal_equations(neural_points, geo_decoder, config, rot, pos, pc_imu):
    """(S [18,18] f64 = H^T R_inv H, HtRz [18] f64 = H^T R_inv z, n_valid) in one launch."""
    x, _, ne = _launch(neural_points, geo_decoder, config, rot, pos, pc_imu, False, True)
    ne = ne.sum(0)  # the kernel spreads its float64 atomics over 16 line-separated copies
    S = torch.zeros((18, 18), device=x.device, dtype=torch.float64)
    iu = torch.triu_indices(6, 6, device=x.device)
    S[iu[0], iu[1]] = ne[:21]
    S[iu[1], iu[0]] = ne[:21]
    b = torch.zeros(18, device=x.device, dtype=torch.float64)
    b[:6] = ne[21:27]
    return S, b, int(ne[27].item())  # (the caller solves the 6 x 6 system on the host next: this read-back is its input)


class IEKFOMMeasurement:
    """`IEKFOM.h_model` with the reference's METHOD signature (utils/error_state_iekf.py:176): a mixin / patch for the
    reference's filter class, which keeps everything else (predict, boxplus, covariance algebra: host-side 18 x 18 math,
    out of scope).  It reads what the reference method reads of `self` -- `neural_points`, `geo_decoder`, `config`,
    `x.rot`, `x.pos` -- stores `self.R_inv` and returns `(sdf_residual, H, valid_points)`:

        from utils.error_state_iekf import IEKFOM
        IEKFOM.h_model = clid_slam_amd.tracking.IEKFOMMeasurement.h_model          # update_iterated() stays unchanged
    """

    def h_model(self, pc_imu: torch.Tensor):
        z, H, valid_points, r_inv = h_model(self.neural_points, self.geo_decoder, self.config, self.x.rot, self.x.pos, pc_imu)
        self.R_inv = r_inv
        return z, H, valid_points
