"""Synthetic inputs for tests and bench (datasets are not shipped with the reference).

"Box room" scene of SURVEY.md section 8(d): an Ouster-128-like scan taken from inside an
axis-aligned box, ray samples with projective SDF labels in the layout the reference's sampler
produces (8 samples per ray, ray-major; `utils/data_sampler.py:16-258`), all on the CPU with a
seeded generator so the CPU and GPU runs see identical inputs.  This is input synthesis only: it is
not part of the hot path and is never timed.
"""
from __future__ import annotations

import math

import torch


def box_room_scan(
    n_elev: int = 128,
    n_azim: int = 1024,
    seed: int = 42,
    sensor=(0.0, 0.0, 1.5),
    box_min=(-20.0, -15.0, 0.0),
    box_max=(20.0, 15.0, 8.0),
    noise_std: float = 0.01,
    min_range: float = 1.0,
    max_range: float = 60.0,
    vox_down_m: float = 0.1,
) -> torch.Tensor:
    """Return scan points [R,3] (fp32) in the sensor frame (sensor frame == world frame shifted by
    ``sensor``; the caller adds the translation back)."""
    gen = torch.Generator().manual_seed(seed)
    elev = torch.linspace(-22.5, 22.5, n_elev, dtype=torch.float64) * math.pi / 180.0
    azim = torch.arange(n_azim, dtype=torch.float64) * (2.0 * math.pi / n_azim)
    ce, se = torch.cos(elev)[:, None], torch.sin(elev)[:, None]
    d = torch.stack(
        (ce * torch.cos(azim)[None], ce * torch.sin(azim)[None], se.expand(-1, n_azim)), dim=-1
    ).reshape(-1, 3)
    o = torch.tensor(sensor, dtype=torch.float64)
    lo = torch.tensor(box_min, dtype=torch.float64)
    hi = torch.tensor(box_max, dtype=torch.float64)
    # distance to the wall each ray leaves through (sensor is inside the box)
    t_hi = (hi - o) / d
    t_lo = (lo - o) / d
    t_axis = torch.where(d > 0, t_hi, t_lo)
    t_axis = torch.where(d == 0, torch.full_like(t_axis, float("inf")), t_axis)
    t = t_axis.min(dim=1).values
    t = t + noise_std * torch.randn(t.shape, generator=gen, dtype=torch.float64)
    pts = (d * t[:, None]).to(torch.float32)
    r = pts.norm(dim=1)
    pts = pts[(r > min_range) & (r < max_range)]
    if vox_down_m > 0:
        key = torch.floor(pts / vox_down_m).to(torch.int64)
        key = key - key.min(dim=0).values
        ext = key.max(dim=0).values + 1
        flat = (key[:, 0] * ext[1] + key[:, 1]) * ext[2] + key[:, 2]
        order = torch.argsort(flat, stable=True)
        sf = flat[order]
        first = torch.ones_like(sf, dtype=torch.bool)
        first[1:] = sf[1:] != sf[:-1]
        pts = pts[order[first].sort().values]
    return pts.contiguous()


def ray_samples(points: torch.Tensor, cfg, seed: int = 43):
    """8 samples / ray with projective labels, ray-major order.

    Layout and ranges follow `utils/data_sampler.py:16-258` (1 endpoint, ``surface_sample_n``
    Gaussian close-to-surface, ``free_front_n`` uniform in front, ``free_behind_n`` uniform behind;
    label = -displacement; surface weights 1 + s/2 - s*r/max_range, free-space weights negated).
    Returns (coord [S,3], sdf_label [S], weight [S]) in the sensor frame.
    """
    gen = torch.Generator().manual_seed(seed)
    R = points.shape[0]
    dist = points.norm(dim=1, keepdim=True)
    ns, nf, nb = cfg.surface_sample_n, cfg.free_front_n, cfg.free_behind_n
    sr = cfg.surface_sample_range_m
    disp = [torch.zeros(R, 1)]
    disp.append(torch.randn(R, ns, generator=gen) * sr)
    front_lo = cfg.free_sample_begin_ratio
    front_hi = 1.0 - 2.0 * sr / dist
    ratio_f = torch.rand(R, nf, generator=gen) * (front_hi - front_lo) + front_lo
    disp.append((ratio_f - 1.0) * dist)
    back_lo = 1.0 + 2.0 * sr / dist
    back_hi = 1.0 + cfg.free_sample_end_dist_m / dist
    ratio_b = torch.rand(R, nb, generator=gen) * (back_hi - back_lo) + back_lo
    disp.append((ratio_b - 1.0) * dist)
    disp = torch.cat(disp, dim=1)  # [R, 8]
    ratio = disp / dist + 1.0
    coord = (points[:, None, :] * ratio[:, :, None]).reshape(-1, 3)
    label = (-disp).reshape(-1)
    w = torch.ones_like(disp)
    w[:, : ns + 1] = 1.0 + cfg.dist_weight_scale * 0.5 - (dist / cfg.max_range) * cfg.dist_weight_scale
    w[:, ns + 1 :] *= -1.0
    return coord.contiguous(), label.contiguous(), w.reshape(-1).contiguous()


def box_room_pool(cfg, n_elev: int = 128, n_azim: int = 1024, seed: int = 42, sensor=(0.0, 0.0, 1.5)):
    """Scan + samples in the WORLD frame.  Returns dict(points, coord, sdf_label, weight, sensor)."""
    pts = box_room_scan(
        n_elev, n_azim, seed, sensor=sensor, min_range=cfg.min_range, max_range=cfg.max_range,
        vox_down_m=cfg.vox_down_m,
    )
    coord, label, weight = ray_samples(pts, cfg, seed + 1)
    off = torch.tensor(sensor, dtype=torch.float32)
    return {
        "points": pts + off,
        "coord": coord + off,
        "sdf_label": label,
        "weight": weight,
        "sensor": off,
    }


# ---------------------------------------------------------------------------------------------------------------
# Sequence of BASELINE.json configs[4] / SURVEY.md section 8(d): the datasets are not shipped, so the "full
# sequence" is a synthetic sweep through a large hall -- 1 m per frame, a straight leg, one 90 degree turn, a
# second leg -- seen by the same Ouster-128 pattern.  Input synthesis only; never timed.
HALL_MIN = (-45.0, -45.0, 0.0)
HALL_MAX = (170.0, 130.0, 8.0)


def sweep_poses(n_frames: int = 200, step_m: float = 1.0, turn_at: int = 120, height: float = 1.5) -> torch.Tensor:
    """[n_frames, 4, 4] float64 sensor->world poses: +x for `turn_at` frames, then yaw 90 degrees and +y."""
    poses = torch.eye(4, dtype=torch.float64).repeat(n_frames, 1, 1)
    x = y = 0.0
    for i in range(n_frames):
        yaw = 0.0 if i < turn_at else math.pi / 2
        if i > 0:
            if i <= turn_at:
                x += step_m
            else:
                y += step_m
        c, s = math.cos(yaw), math.sin(yaw)
        poses[i, :3, :3] = torch.tensor([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]], dtype=torch.float64)
        poses[i, :3, 3] = torch.tensor([x, y, height], dtype=torch.float64)
    return poses


def hall_scan(pose: torch.Tensor, seed: int, device="cpu", n_elev: int = 128, n_azim: int = 1024, noise_std: float = 0.01,
              min_range: float = 1.0, max_range: float = 60.0, vox_down_m: float = 0.1,
              box_min=HALL_MIN, box_max=HALL_MAX) -> torch.Tensor:
    """Scan points [R, 3] fp32 in the SENSOR frame of `pose` (4x4 sensor->world) inside the axis-aligned hall."""
    gen = torch.Generator(device=device).manual_seed(seed)
    elev = torch.linspace(-22.5, 22.5, n_elev, dtype=torch.float64, device=device) * math.pi / 180.0
    azim = torch.arange(n_azim, dtype=torch.float64, device=device) * (2.0 * math.pi / n_azim)
    ce, se = torch.cos(elev)[:, None], torch.sin(elev)[:, None]
    d_s = torch.stack((ce * torch.cos(azim)[None], ce * torch.sin(azim)[None], se.expand(-1, n_azim)), dim=-1).reshape(-1, 3)
    pose = pose.to(device=device, dtype=torch.float64)
    d = d_s @ pose[:3, :3].T  # world-frame ray directions
    o = pose[:3, 3]
    lo = torch.tensor(box_min, dtype=torch.float64, device=device)
    hi = torch.tensor(box_max, dtype=torch.float64, device=device)
    t_axis = torch.where(d > 0, (hi - o) / d, (lo - o) / d)
    t_axis = torch.where(d == 0, torch.full_like(t_axis, float("inf")), t_axis)
    t = t_axis.min(dim=1).values
    t = t + noise_std * torch.randn(t.shape, generator=gen, dtype=torch.float64, device=device)
    pts = (d_s * t[:, None]).to(torch.float32)
    r = pts.norm(dim=1)
    pts = pts[(r > min_range) & (r < max_range)]
    if vox_down_m > 0:
        key = torch.floor(pts / vox_down_m).to(torch.int64)
        key = key - key.min(dim=0).values
        ext = key.max(dim=0).values + 1
        flat = (key[:, 0] * ext[1] + key[:, 1]) * ext[2] + key[:, 2]
        order = torch.argsort(flat, stable=True)
        sf = flat[order]
        first = torch.ones_like(sf, dtype=torch.bool)
        first[1:] = sf[1:] != sf[:-1]
        pts = pts[order[first].sort().values]
    return pts.contiguous()
