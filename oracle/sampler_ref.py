"""CPU ORACLE for the step in front of the hot path ("next" row N2 of SURVEY.md section 8f): the per-frame
sample + label generation that fills the training pool.  TEST INFRASTRUCTURE ONLY (same rules as
oracle/cpu_ref.py: only tests/, smoke() and bench legs may import it; the product never does).

Restated from the reference (paths relative to the reference root):
  * the raw-point voxel map and its region-specific SDF estimate   model/local_point_cloud_map.py:11-201
  * the ray sampler (projective labels and the region-specific variant)  utils/data_sampler.py:16-402
  * the pool bookkeeping of Mapper.process_frame                    utils/mapper.py:159-470
  * voxel down-sampling                                             utils/tools.py:639-682

Parity status: PINNED against outputs of the reference itself (fixture G9, oracle/make_golden.py --only-g9,
checked by tests/test_oracle_golden.py).  Random draws are an INPUT here (`noise`): the reference draws them
from torch's global generator in a fixed order (randn [R*n_surf,1], rand [R*n_front,1], rand [R*n_behind,1]).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Optional

import torch

CLOUD_PRIMES = (73856093, 19349663, 83492791)  # model/local_point_cloud_map.py:27-29 (NOT the neural-point primes)


def voxel_down_sample(points: torch.Tensor, voxel_size: float) -> torch.Tensor:
    """utils/tools.py:639-682: per voxel the index of the point closest to the voxel centre (distance
    quantised to 1000 levels, lowest index among equals), voxels in ascending linear order.  The linear
    voxel id uses stride max(cell) (not max+1), as the reference does."""
    cell_f = torch.floor(points / voxel_size)
    d = ((points - (cell_f + 0.5) * voxel_size) ** 2).sum(dim=1) ** 0.5
    q = (d / d.max() * 999).long()
    cell = cell_f.long() - torch.floor(points.min(dim=0)[0] / voxel_size).long()
    stride = cell.max()
    lin = cell[:, 0] + cell[:, 1] * stride + cell[:, 2] * stride * stride
    n = points.shape[0]
    big = 10 ** len(str(n - 1))
    key = torch.arange(n) + q * big
    _, inv = torch.unique(lin, return_inverse=True)
    best = torch.full((int(inv.max()) + 1,), torch.iinfo(torch.int64).max, dtype=torch.int64)
    best.scatter_reduce_(0, inv, key, reduce="amin", include_self=True)
    return best % big


@dataclass
class LocalCloud:
    """model/local_point_cloud_map.py:11-36."""

    buffer_pt_index: torch.Tensor  # [B] int64, -1 = empty
    points: torch.Tensor  # [M,3] f32, world frame
    resolution: float = 0.2
    buffer_size: int = int(5e6)
    map_size: float = 100.0
    neighbor_idx: torch.Tensor = None  # [P,3] int64
    max_valid_range: float = 0.0

    @staticmethod
    def empty(resolution=0.2, buffer_size=int(5e6), map_size=100.0, num_nei_cells=1, search_alpha=0.2):
        lc = LocalCloud(torch.full((buffer_size,), -1, dtype=torch.int64), torch.empty((0, 3)), resolution,
                        buffer_size, map_size)
        lc.neighbor_idx, lc.max_valid_range = cloud_neighborhood(num_nei_cells, search_alpha, resolution)
        return lc


def cloud_neighborhood(num_nei_cells: int, search_alpha: float, resolution: float):
    """model/local_point_cloud_map.py:74-96: cells inside the sphere of radius (n + alpha), and the
    "no neighbour" distance 1.732 (n + 1) r."""
    r = torch.arange(-num_nei_cells, num_nei_cells + 1, dtype=torch.int64)
    dx = torch.stack(torch.meshgrid(r, r, r, indexing="ij"), dim=-1).reshape(-1, 3)
    keep = (dx**2).sum(-1) < (num_nei_cells + search_alpha) ** 2
    return dx[keep], 1.732 * (num_nei_cells + 1) * resolution


def cloud_hash(lc: LocalCloud, cells: torch.Tensor) -> torch.Tensor:
    """:38-41.  fmod keeps the sign; the table is then indexed python-style, i.e. slot h + B for h < 0."""
    primes = torch.tensor(CLOUD_PRIMES, dtype=torch.int64)
    return torch.fmod((cells * primes).sum(-1), lc.buffer_size)


def cloud_insert(lc: LocalCloud, points: torch.Tensor) -> None:
    """:43-61: voxel-down-sampled points whose slot is empty are appended (later duplicates of a slot
    within one call overwrite earlier ones)."""
    p = points[voxel_down_sample(points, lc.resolution)]
    h = cloud_hash(lc, torch.floor(p / lc.resolution).long())
    free = lc.buffer_pt_index[h] == -1
    fresh = p[free]
    lc.buffer_pt_index[h[free]] = torch.arange(fresh.shape[0]) + lc.points.shape[0]
    lc.points = torch.cat((lc.points, fresh), 0)


def cloud_update(lc: LocalCloud, sensor_position: torch.Tensor, points: torch.Tensor) -> None:
    """:63-72: insert, drop everything farther than map_size from the sensor, rebuild the table."""
    cloud_insert(lc, points)
    keep = torch.norm(lc.points - sensor_position, dim=-1) < lc.map_size
    lc.points = lc.points[keep]
    tab = torch.full((lc.buffer_size,), -1, dtype=torch.int64)
    tab[cloud_hash(lc, torch.floor(lc.points / lc.resolution).long())] = torch.arange(lc.points.shape[0])
    lc.buffer_pt_index = tab


def fit_planes(knn: torch.Tensor, eta_threshold: float = 0.2, threshold: float = 0.1):
    """model/local_point_cloud_map.py:156-201: total-least-squares plane through each group of points
    (normal = right singular vector of the smallest singular value of the centred points), accepted when
    s_min / (s_mid + 1e-6) <= eta_threshold and every point lies within `threshold` of the plane."""
    c = knn.mean(dim=1, keepdim=True)
    _, s, vh = torch.linalg.svd(knn - c, full_matrices=False)
    flat = s[:, -1] / (s[:, 1] + 1e-6) <= eta_threshold
    normal = torch.zeros((knn.shape[0], 3), dtype=knn.dtype)
    normal[flat] = vh[:, -1, :][flat]
    offset = -1.0 * torch.sum(normal * c.squeeze(1), dim=1)
    resid = torch.abs(torch.bmm(knn, normal.unsqueeze(-1)).squeeze(-1) + offset.unsqueeze(-1))
    ok = (resid.max(dim=1).values <= threshold) & flat
    return normal, offset, ok


def region_sdf(lc: LocalCloud, points: torch.Tensor):
    """model/local_point_cloud_map.py:98-153: |SDF| label of each (world-frame) sample from the raw points
    around it: the distance to the plane through its 4 nearest map points when that plane is trustworthy,
    else the distance to the nearest one; `surface_mask` = at least one map point in the neighbourhood."""
    n = points.shape[0]
    sdf_abs = torch.full((n,), lc.max_valid_range)
    surface = torch.ones(n, dtype=torch.bool)
    step = 262144
    for a in range(0, n, step):
        p = points[a:a + step]
        cells = torch.floor(p / lc.resolution).long()[:, None, :] + lc.neighbor_idx
        idx = lc.buffer_pt_index[cloud_hash(lc, cells)]
        nb = lc.points[idx]
        d = torch.norm(nb - p[:, None, :], dim=-1)
        d = torch.where(idx == -1, torch.tensor(lc.max_valid_range, dtype=d.dtype), d)
        d4, i4 = torch.topk(d, 4, largest=False, dim=1)
        knn = torch.gather(nb, 1, i4.unsqueeze(-1).expand(-1, -1, 3))
        four = d4[:, 3] < lc.max_valid_range
        normal = torch.zeros_like(p)
        offset = torch.zeros(p.shape[0])
        ok = torch.zeros(p.shape[0], dtype=torch.bool)
        nv, ov, okv = fit_planes(knn[four])
        normal[four], offset[four], ok[four] = nv, ov, okv
        plane = torch.abs(torch.sum(normal * p, dim=1) + offset)
        sdf_abs[a:a + step] = torch.where(ok, plane, d4[:, 0])
        surface[a:a + step] = d4[:, 0] < lc.max_valid_range
    return sdf_abs, surface


def region_sdf_ambiguity(lc: LocalCloud, points: torch.Tensor, coord_tol: float = 2e-5, eta_tol: float = 2e-3,
                         resid_tol: float = 2e-4, tie_tol: float = 1e-5, label_tol: float = 5e-5, noise: float = 6e-6):
    """Checker aid (no reference counterpart): the samples whose label / surface mask from `region_sdf`
    (model/local_point_cloud_map.py:98-201) is NOT a continuous function of fp32 rounding, so that two correct
    implementations -- different summation orders in the rigid transform, the centring and the 4 x 3 SVD, on world coordinates
    of tens of metres -- may legitimately disagree by more than rounding.  Returns a dict of bool masks [n]:
      cell       a coordinate within `coord_tol` of a voxel boundary (floor(p / r) may differ: another neighbourhood)
      range      the nearest or the 4th-nearest distance within `coord_tol` of the "no neighbour" range (mask / fit validity)
      tie        4th and 5th nearest neighbour within `tie_tol` of each other (another set of 4 points -> another plane)
      eta        eta = s_min / (s_mid + 1e-6) within `eta_tol` of its threshold 0.2 (plane accepted by one side only)
      resid      largest point-to-plane residual within `resid_tol` of its threshold 0.1 (likewise)
      illcond    accepted plane whose normal is badly conditioned: the label's sensitivity to `noise` metres of centring
                 error, |p - c| * noise / (s_mid - s_min), exceeds `label_tol` (nearly collinear points of one scan line)
      any        the union.
    A checker holds every sample OUTSIDE `any` to the strict label tolerance and an exact surface mask."""
    n = points.shape[0]
    out = {k: torch.zeros(n, dtype=torch.bool) for k in ("cell", "range", "tie", "eta", "resid", "illcond")}
    step = 262144
    for a in range(0, n, step):
        p = points[a:a + step]
        sl = slice(a, a + p.shape[0])
        u = p / lc.resolution
        out["cell"][sl] = ((u - torch.round(u)).abs() * lc.resolution < coord_tol).any(dim=1)
        cells = torch.floor(u).long()[:, None, :] + lc.neighbor_idx
        idx = lc.buffer_pt_index[cloud_hash(lc, cells)]
        nb = lc.points[idx]
        d = torch.norm(nb - p[:, None, :], dim=-1)
        d = torch.where(idx == -1, torch.tensor(lc.max_valid_range, dtype=d.dtype), d)
        kk = min(5, d.shape[1])
        d5, i5 = torch.topk(d, kk, largest=False, dim=1)
        real_near = (idx != -1) & ((d - lc.max_valid_range).abs() < coord_tol)   # a REAL distance at the sentinel's value
        out["range"][sl] = real_near.any(dim=1)
        four = d5[:, 3] < lc.max_valid_range
        if kk == 5:
            out["tie"][sl] = four & ((d5[:, 4] - d5[:, 3]) < tie_tol)
        knn = torch.gather(nb, 1, i5[:, :4].unsqueeze(-1).expand(-1, -1, 3))[four]
        c = knn.mean(dim=1, keepdim=True)
        _, sv, vh = torch.linalg.svd(knn - c, full_matrices=False)
        eta = sv[:, -1] / (sv[:, 1] + 1e-6)
        flat = eta <= 0.2
        normal = vh[:, -1, :]
        offset = -1.0 * torch.sum(normal * c.squeeze(1), dim=1)
        resid = torch.abs(torch.bmm(knn, normal.unsqueeze(-1)).squeeze(-1) + offset.unsqueeze(-1)).max(dim=1).values
        lever = torch.norm(p[four] - c.squeeze(1), dim=1)
        sens = lever * noise / (sv[:, 1] - sv[:, -1]).clamp_min(1e-12)
        ok = flat & (resid <= 0.1)
        fi = torch.nonzero(four).flatten() + a
        out["eta"][fi] = (eta - 0.2).abs() < eta_tol
        out["resid"][fi] = flat & ((resid - 0.1).abs() < resid_tol)
        out["illcond"][fi] = ok & (sens > label_tol)
    out["any"] = out["cell"] | out["range"] | out["tie"] | out["eta"] | out["resid"] | out["illcond"]
    return out


def transform(points: torch.Tensor, pose: torch.Tensor) -> torch.Tensor:
    """utils/tools.py:590-609 (homogeneous row-vector product in the points' dtype)."""
    homo = torch.cat([points, torch.ones(points.shape[0], 1).to(points)], dim=1)
    return torch.matmul(homo, pose.to(points).T)[:, :3]


@dataclass
class SamplerConfig:
    """The sampler's keys of utils/config.py (values of config/run_ncd128.yaml)."""

    surface_sample_range_m: float = 0.25
    surface_sample_n: int = 4
    free_behind_n: int = 1
    free_front_n: int = 2
    free_sample_begin_ratio: float = 0.5
    free_sample_end_dist_m: float = 1.2
    dist_weight_on: bool = True
    dist_weight_scale: float = 0.8
    behind_dropoff_on: bool = False
    max_range: float = 60.0


def _ray_samples(cfg: SamplerConfig, pts: torch.Tensor, noise):
    """The part both samplers share (utils/data_sampler.py:36-134 == :275-338): per ray 1 exact sample,
    n_surf Gaussian samples around the hit, n_front uniform samples in front, n_behind behind; returned
    sample-type-major (all rays' exact samples, then all rays' 1st surface sample, ...)."""
    R = pts.shape[0]
    ns, nf, nb = cfg.surface_sample_n, cfg.free_front_n, cfg.free_behind_n
    z_s, u_f, u_b = noise
    assert z_s.shape == (R * ns, 1) and u_f.shape == (R * nf, 1) and u_b.shape == (R * nb, 1)
    dist = torch.linalg.norm(pts, dim=1, keepdim=True)
    sig = cfg.surface_sample_range_m
    disp_s = z_s * sig
    ratio_s = disp_s / dist.repeat(ns, 1) + 1.0
    df = dist.repeat(nf, 1)
    ratio_f = u_f * ((1.0 - 2.0 * sig / df) - cfg.free_sample_begin_ratio) + cfg.free_sample_begin_ratio
    disp_f = (ratio_f - 1.0) * df
    db = dist.repeat(nb, 1)
    lo_b = 1.0 + 2.0 * sig / db
    ratio_b = u_b * ((cfg.free_sample_end_dist_m / db + 1.0) - lo_b) + lo_b
    disp_b = (ratio_b - 1.0) * db
    disp = torch.cat((torch.zeros_like(dist), disp_s, disp_f, disp_b), 0)
    ratio = torch.cat((torch.ones_like(dist), ratio_s, ratio_f, ratio_b), 0)
    n_all = 1 + ns + nf + nb
    xyz = pts.repeat(n_all, 1) * ratio
    depth_of_ray = dist.repeat(n_all, 1)
    return xyz, disp, ratio, depth_of_ray, disp_s, n_all


def _weights(cfg: SamplerConfig, depth_of_ray, ratio, R, n_all):
    w = torch.ones_like(depth_of_ray * ratio)
    n_surface = R * (cfg.surface_sample_n + 1)
    if cfg.dist_weight_on:  # utils/data_sampler.py:143-152 / :372-381
        w[:n_surface] = 1 + cfg.dist_weight_scale * 0.5 - (depth_of_ray[:n_surface] / cfg.max_range) * cfg.dist_weight_scale
    return w, n_surface


def _ray_major(t: torch.Tensor, n_all: int):
    """utils/data_sampler.py:208-216 / :386-400: sample-type-major -> ray-major."""
    if t.dim() == 2 and t.shape[1] == 3:
        return t.reshape(n_all, -1, 3).transpose(0, 1).reshape(-1, 3)
    return t.reshape(n_all, -1).transpose(0, 1).reshape(-1)


def sample_projective(cfg: SamplerConfig, pts: torch.Tensor, noise):
    """DataSampler.sample_pin (utils/data_sampler.py:16-258), geometry only: (coord [R*n_all,3] sensor frame,
    sdf_label, weight); label = -displacement along the ray, weight < 0 marks free-space samples."""
    xyz, disp, ratio, depth, _, n_all = _ray_samples(cfg, pts, noise)
    w, n_surface = _weights(cfg, depth, ratio, pts.shape[0], n_all)
    if cfg.behind_dropoff_on:  # :154-164
        hi = cfg.free_sample_end_dist_m
        lo = 0.2 * hi
        w = w * (torch.clamp((hi - disp) / (hi - lo), min=0.0, max=1.0) * 0.8 + 0.2)
    w[n_surface:] *= -1.0
    label = _ray_major(disp.squeeze(1), n_all) * -1
    return _ray_major(xyz, n_all), label, _ray_major(w.squeeze(1), n_all)


def sample_region_specific(cfg: SamplerConfig, pts: torch.Tensor, lc: LocalCloud, pose: torch.Tensor, noise):
    """DataSampler.sample (utils/data_sampler.py:260-402): as above, but the near-surface samples take
    sign(-displacement) x the region-specific |SDF| of their world position, and the ones with no raw map
    point around them are dropped."""
    R = pts.shape[0]
    xyz, disp, ratio, depth, disp_s, n_all = _ray_samples(cfg, pts, noise)
    sign = torch.where(disp_s.squeeze(1) < 0, 1, -1)
    keep = torch.ones(R * n_all, dtype=torch.bool)
    label = -1 * disp.squeeze(1)
    n_surface = R * (cfg.surface_sample_n + 1)
    d, ok = region_sdf(lc, transform(xyz[R:n_surface], pose))
    keep[R:n_surface] = ok
    label[R:n_surface] = sign * d
    w, _ = _weights(cfg, depth, ratio, R, n_all)
    w[n_surface:] *= -1.0
    keep = _ray_major(keep, n_all)
    return _ray_major(xyz, n_all)[keep], _ray_major(label, n_all)[keep], _ray_major(w.squeeze(1), n_all)[keep]


@dataclass
class PoolState:
    """The training pool of utils/mapper.py:84-97."""

    coord: torch.Tensor
    global_coord: torch.Tensor
    sdf_label: torch.Tensor
    weight: torch.Tensor
    time: torch.Tensor
    cur_sample_count: int = 0
    pool_sample_count: int = 0

    @staticmethod
    def empty():
        return PoolState(torch.empty((0, 3)), torch.empty((0, 3)), torch.empty(0), torch.empty(0),
                         torch.empty(0, dtype=torch.int32))


def pool_append_and_filter(pool: PoolState, coord, sdf_label, weight, frame_id: int, pose: torch.Tensor,
                           window_radius: float, pool_capacity: int, discard: Optional[torch.Tensor] = None):
    """utils/mapper.py:286-392 (no bundle adjustment, pool_filter_freq = 1): append this frame's samples
    (sensor frame + world frame), keep what lies inside the window around the sensor; beyond the capacity
    `discard` (random positions among the kept ones, drawn by the caller) are dropped as well."""
    n = coord.shape[0]
    pool.coord = torch.cat((pool.coord, coord), 0)
    pool.weight = torch.cat((pool.weight, weight), 0)
    pool.sdf_label = torch.cat((pool.sdf_label, sdf_label), 0)
    pool.time = torch.cat((pool.time, torch.full((n,), frame_id, dtype=torch.int32)), 0)
    pool.global_coord = torch.cat((pool.global_coord, transform(coord, pose)), 0)
    # the pose is float64 in the reference, so this test is evaluated in float64 (type promotion)
    keep = ((pool.global_coord - pose[:3, 3]) ** 2).sum(-1) < window_radius**2
    kept = torch.nonzero(keep).squeeze()
    if kept.shape[0] > pool_capacity:
        assert discard is not None and discard.shape[0] == kept.shape[0] - pool_capacity
        keep[kept[discard]] = False
    for f in ("coord", "global_coord", "sdf_label", "weight", "time"):
        setattr(pool, f, getattr(pool, f)[keep])
    pool.cur_sample_count = int(keep[-n:].sum())
    pool.pool_sample_count = int(keep.sum())
    return keep


def adaptive_iter_offset(new_ratio: float, frame_id: int, adaptive: bool = True, less=0.02, more=0.15,
                         restart=0.3, freeze_after_frame=40) -> int:
    """utils/mapper.py:448-462."""
    if not adaptive:
        return 0
    if new_ratio < less:
        return -5
    if new_ratio > more:
        return 10 if (frame_id > freeze_after_frame and new_ratio > restart) else 5
    return 0
