"""CPU ORACLE for CLID-SLAM's per-scan SDF training inner loop.  TEST INFRASTRUCTURE ONLY.

This file is a from-scratch CPU (PyTorch fp32, eager) restatement of the reference algorithm on
the hot path named by BASELINE.json `north_star`.  Only `tests/`, `__graft_entry__.smoke()` and
`bench.py`'s `cpu_baseline` leg may import it; the product path (`clid-slam_amd/`) never does and
fails loudly when the HIP library is missing.

Parity status: PINNED.  The reference ships no tests or golden vectors, so
the restatement is pinned against outputs of the reference itself, imported unmodified in the
build container by `oracle/make_golden.py` (fixtures G1..G6 under `tests/golden/`, checked by
`tests/test_oracle_golden.py`).

Every function cites the reference lines it follows (paths relative to the reference root).
State is held in a plain `MapState`; nothing here depends on the product package.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Optional

import torch
import torch.nn.functional as F

PRIMES = (73856093, 19349669, 83492791)  # model/neural_points.py:79-81


# --------------------------------------------------------------------------------------------------
# state
# --------------------------------------------------------------------------------------------------
@dataclass
class MapState:
    """The tensors the hot path reads/writes (model/neural_points.py:79-133)."""

    buffer_pt_index: torch.Tensor  # [B] int64, -1 = empty slot
    neural_points: torch.Tensor  # [Mg,3] f32
    point_ts_create: torch.Tensor  # [Mg] int32
    travel_dist: torch.Tensor  # [frames] f32
    cur_ts: int
    global2local: torch.Tensor  # [Mg+1] int64 (last = -1)
    local_neural_points: torch.Tensor  # [M,3]
    local_geo_features: torch.Tensor  # [M+1,F] (last row = padding)
    local_point_certainties: torch.Tensor  # [M]
    local_point_ts_update: torch.Tensor  # [M] int32
    # global-map counterparts (query_locally=False)
    geo_features: Optional[torch.Tensor] = None  # [Mg+1,F]
    point_certainties: Optional[torch.Tensor] = None  # [Mg]
    # search parameters
    resolution: float = 0.4
    buffer_size: int = int(5e7)
    diff_travel_dist_local: float = 310.0
    temporal_local_map_on: bool = True
    neighbor_dx: torch.Tensor = field(default=None)  # [P,3] int64
    max_valid_dist2: float = 4.32
    # query parameters
    nn_k: int = 6
    layer_norm_on: bool = False
    weighted_first: bool = True


@dataclass
class DecoderParams:
    """model/decoder.py:12-56 with hidden_level == 1: Linear(D,H) -> ReLU -> Linear(H,1)."""

    W1: torch.Tensor  # [H,D]
    b1: torch.Tensor  # [H]
    W2: torch.Tensor  # [1,H]
    b2: torch.Tensor  # [1]
    sdf_scale: float = 0.055

    def tensors(self):
        return [self.W1, self.b1, self.W2, self.b2]


# --------------------------------------------------------------------------------------------------
# a2: neighbourhood search
# --------------------------------------------------------------------------------------------------
def search_neighborhood(num_nei_cells: int, search_alpha: float, resolution: float):
    """model/neural_points.py:931-969 -> (neighbor_dx [P,3] int64, max_valid_dist2)."""
    r = torch.arange(-num_nei_cells, num_nei_cells + 1, dtype=torch.int64)
    gx, gy, gz = torch.meshgrid(r, r, r, indexing="ij")
    cube = torch.stack((gx, gy, gz), dim=-1).reshape(-1, 3)
    keep = (cube * cube).sum(-1) < (num_nei_cells + search_alpha) ** 2
    return cube[keep].contiguous(), 3 * ((num_nei_cells + 1) * resolution) ** 2


def radius_neighborhood_search(st: MapState, points: torch.Tensor, time_filtering: bool = False):
    """model/neural_points.py:971-1030 -> (dist2 [N,P] f32, neighb_idx [N,P] int64, global ids).

    Voxel index = floor(x / res) with a true fp32 divide; the hash
    `fmod(sum(cell * primes), B)` indexed with Python negative wrap-around is the non-negative
    modulo.  Index -1 reads the LAST element of the indexed arrays and is masked afterwards.
    """
    primes = torch.tensor(PRIMES, dtype=torch.int64)
    cell = torch.floor(points / st.resolution).to(torch.int64)  # [N,3]
    cells = cell[:, None, :] + st.neighbor_dx[None, :, :]  # [N,P,3]
    slot = torch.remainder((cells * primes).sum(-1), int(st.buffer_size))
    idx = st.buffer_pt_index[slot]  # [N,P]
    if time_filtering:
        created = st.point_ts_create[idx]
        gap = torch.abs(st.travel_dist[st.cur_ts] - st.travel_dist[created])
        idx = torch.where(gap < st.diff_travel_dist_local, idx, torch.full_like(idx, -1))
    diff = st.neural_points[idx] - points[:, None, :]
    dist2 = (diff * diff).sum(-1)
    dist2 = torch.where(idx == -1, torch.full_like(dist2, st.max_valid_dist2), dist2)
    idx = torch.where(dist2 > st.max_valid_dist2, torch.full_like(idx, -1), idx)
    return dist2, idx


# --------------------------------------------------------------------------------------------------
# a3: feature query
# --------------------------------------------------------------------------------------------------
def query_feature(
    st: MapState,
    query_points: torch.Tensor,
    query_ts: Optional[torch.Tensor] = None,
    training_mode: bool = True,
    query_locally: bool = True,
):
    """model/neural_points.py:553-769 (geometry features only, no colour, no PGO rotation).

    Returns (geo_feat [N,D] or [N,K,D], weight [N,K,1], nn_counts [N], certainty [N],
             idx [N,K] (local or global ids, -1 invalid)).
    TIES: the reference sorts with torch.sort's default (unstable) algorithm, so which of several EQUIDISTANT candidates
    takes the K-th place is implementation-defined there (CPU and CUDA differ, and so do row lengths).  The restatement
    fixes the choice to the stable order -- the candidate with the lower probe index wins -- which is also the rule of the
    HIP kernels (csrc/train.hip search8, csrc/common.hpp search_topk); on the reference's own golden vectors (no
    equidistant pair of different points at the cut) both orders give the same result.
    Differentiable w.r.t. `query_points` (through r_k and the IDW weights) and the feature table.
    """
    K = st.nn_k
    dist2, idx = radius_neighborhood_search(
        st, query_points, time_filtering=st.temporal_local_map_on and query_locally
    )
    if query_locally:
        idx = st.global2local[idx]  # :595-598
        feats_tab, pts_tab, cert_tab = st.local_geo_features, st.local_neural_points, st.local_point_certainties
    else:
        feats_tab, pts_tab, cert_tab = st.geo_features, st.neural_points, st.point_certainties
    nn_counts = (idx >= 0).sum(-1)  # :600-602 (over all P probes)
    dist2 = torch.where(idx == -1, torch.full_like(dist2, 9e3), dist2)  # :606
    dist2, order = torch.sort(dist2, dim=1, stable=True)  # :607-609 (equal distances: see TIES below)
    idx = idx.gather(1, order)[:, :K]
    dist2 = dist2[:, :K]
    valid = idx >= 0
    validf = valid.unsqueeze(-1)

    feat = torch.where(validf, feats_tab[idx], torch.zeros((), dtype=feats_tab.dtype))  # :620-631
    if st.layer_norm_on:
        feat = F.layer_norm(feat, [feat.shape[-1]])  # :632-633 (all-zero rows stay zero)
    cert = torch.where(valid, cert_tab[idx], torch.zeros((), dtype=cert_tab.dtype))  # :654,734
    rel = torch.where(validf, query_points[:, None, :] - pts_tab[idx], torch.zeros(()))  # :655-674
    vec = torch.cat((feat, rel), dim=2)  # :679-682

    eps = 1e-15
    w = torch.where(valid, 1.0 / (dist2 + eps), torch.zeros(()))  # :688-693
    w = torch.where((nn_counts == 0)[:, None], torch.full_like(w, eps), w)  # :694-696
    w = w / w.sum(dim=1, keepdim=True)  # :699-702
    w = torch.where(valid, w, torch.zeros(()))  # :706

    with torch.no_grad():  # :708-741
        if training_mode:
            tgt = torch.where(valid, idx, torch.zeros_like(idx)).flatten()
            cert_tab.scatter_add_(0, tgt, w.detach().flatten())
            if query_locally and query_ts is not None:
                ts_k = torch.where(valid, query_ts.view(-1, 1).expand(-1, K), torch.zeros((), dtype=query_ts.dtype))
                st.local_point_ts_update.scatter_reduce_(
                    0, tgt, ts_k.flatten().to(st.local_point_ts_update.dtype), reduce="amax", include_self=True
                )
        certainty = (cert * w).sum(dim=1)  # uses the pre-update certainties gathered above

    w = w.unsqueeze(-1)
    if st.weighted_first:
        vec = (vec * w).sum(dim=1)  # :745-754
    return vec, w, nn_counts, certainty, idx


def query_certainty(st: MapState, query_points: torch.Tensor):
    """model/neural_points.py:1032-1051 (global certainties, max over the probed cells)."""
    _, idx = radius_neighborhood_search(st, query_points)
    c = torch.where(idx < 0, torch.zeros(()), st.point_certainties[idx])
    return c.max(dim=-1).values


# --------------------------------------------------------------------------------------------------
# a4: decoder
# --------------------------------------------------------------------------------------------------
# Checker aid for BASELINE.json configs[2] (no reference counterpart: the reference has no reduced-precision mode).  None = the
# reference's fp32 decoder.  A dict switches layer 1 of `mlp_sdf` to the arithmetic the bf16 instantiation of the tile decode
# kernel performs (clid-slam_amd/csrc/train_tile.hip PREC = 1), so that the HIP bf16 path is compared with the VALUES bf16
# operand rounding produces -- at fp32 summation-order tolerance -- instead of with the fp32 oracle at a bar widened by a guess:
#   forward   pre = bf16(W1) bf16(f) + b1      operands rounded to nearest-even bf16, products and sums fp32 (the bias rides as
#                                              hi + lo parts: exact to 2^-17)
#   backward  d f = bf16(dh) bf16(W1)          dh = dL/dpre in fp32, rounded as the MFMA operand
#             dW1 = bf16(dh)^T bf16(f), db1 = sum bf16(dh)   ("dW1": True, what the bf16 instantiations do at every launch size;
#                                              False: contracted in fp32 from the unrounded values)
#   layer 2 (64 in-lane FMAs), the loss and everything outside the three contractions stay fp32 on both sides.
BF16_CONTRACTIONS = None


def _bf16(x: torch.Tensor) -> torch.Tensor:
    return x.to(torch.bfloat16).to(x.dtype)


class _Bf16Layer1(torch.autograd.Function):
    @staticmethod
    def forward(ctx, f, W1, b1, round_dw1):
        fb, Wb = _bf16(f), _bf16(W1)
        ctx.save_for_backward(f, fb, Wb)
        ctx.round_dw1 = bool(round_dw1)
        return F.linear(fb, Wb) + b1

    @staticmethod
    def backward(ctx, g):
        f, fb, Wb = ctx.saved_tensors
        gb = _bf16(g)
        df = gb @ Wb
        g2 = (gb if ctx.round_dw1 else g).reshape(-1, g.shape[-1])
        f2 = (fb if ctx.round_dw1 else f).reshape(-1, f.shape[-1])
        return df, g2.t() @ f2, g2.sum(dim=0), None


def mlp_sdf(dec: DecoderParams, features: torch.Tensor) -> torch.Tensor:
    """model/decoder.py:58-82: sdf = scale * (W2 relu(W1 f + b1) + b2), squeezed over dim 1."""
    if BF16_CONTRACTIONS is not None:
        h = F.relu(_Bf16Layer1.apply(features, dec.W1, dec.b1, BF16_CONTRACTIONS.get("dW1", True)))
    else:
        h = F.relu(F.linear(features, dec.W1, dec.b1))
    return F.linear(h, dec.W2, dec.b2).squeeze(1) * dec.sdf_scale


def sdf_at(st: MapState, dec: DecoderParams, x: torch.Tensor) -> torch.Tensor:
    """utils/mapper.py:968-982 (`Mapper.sdf`): default-argument query (training_mode=True,
    query_ts=None, local map) followed by the decoder."""
    f, w, _, _, _ = query_feature(st, x)
    s = mlp_sdf(dec, f)
    if not st.weighted_first:
        s = (s * w).sum(dim=1).squeeze(1)
    return s


# --------------------------------------------------------------------------------------------------
# a6 / a7: gradients of the SDF w.r.t. the query position
# --------------------------------------------------------------------------------------------------
def numerical_gradient(st: MapState, dec: DecoderParams, x: torch.Tensor, eps: float) -> torch.Tensor:
    """utils/mapper.py:985-1034, two-sided: ONE `sdf_at` call on the 6N shifted copies
    [x+ex; x-ex; x+ey; x-ey; x+ez; x-ez]; stays in the autograd graph."""
    n = x.shape[0]
    shifted = []
    for a in range(3):
        e = torch.zeros(3, dtype=x.dtype)
        e[a] = eps
        shifted += [x + e, x - e]
    s = sdf_at(st, dec, torch.cat(shifted, dim=0)).unsqueeze(-1)
    cols = [(s[2 * a * n : (2 * a + 1) * n] - s[(2 * a + 1) * n : (2 * a + 2) * n]) / (2 * eps) for a in range(3)]
    return torch.cat(cols, dim=1)


def autograd_gradient(inputs: torch.Tensor, outputs: torch.Tensor) -> torch.Tensor:
    """utils/tools.py:298-311 (`get_gradient`): d outputs / d inputs with create_graph=True."""
    return torch.autograd.grad(
        outputs, inputs, torch.ones_like(outputs), create_graph=True, retain_graph=True, only_inputs=True
    )[0]


def closed_form_sdf_and_gradient(st: MapState, dec: DecoderParams, x: torch.Tensor):
    """the same forward plus the analytic d sdf / d x written out
    (no autograd).  Used to pin the formula the HIP kernel implements against `autograd_gradient`.
    Returns (sdf [N], grad [N,3], nn_counts [N])."""
    with torch.no_grad():
        K = st.nn_k
        dist2, idx = radius_neighborhood_search(st, x, time_filtering=st.temporal_local_map_on)
        idx = st.global2local[idx]
        nn_counts = (idx >= 0).sum(-1)
        dist2 = torch.where(idx == -1, torch.full_like(dist2, 9e3), dist2)
        dist2, order = torch.sort(dist2, dim=1, stable=True)
        idx = idx.gather(1, order)[:, :K]
        dist2 = dist2[:, :K]
        valid = (idx >= 0).to(x.dtype)
        feat = st.local_geo_features[idx] * valid[..., None]
        if st.layer_norm_on:
            feat = F.layer_norm(feat, [feat.shape[-1]])
        r = (x[:, None, :] - st.local_neural_points[idx]) * valid[..., None]
        omega = valid / (dist2 + 1e-15)
        osum = omega.sum(1, keepdim=True)
        w = torch.where(osum > 0, omega / osum, torch.zeros(()))
        v = torch.cat((feat, r), dim=2)  # [N,K,D]
        f = (v * w[..., None]).sum(1)
        pre = f @ dec.W1.T + dec.b1
        act = (pre > 0).to(x.dtype)
        sdf = dec.sdf_scale * ((pre * act) @ dec.W2[0] + dec.b2[0])
        alpha = 2.0 * r * omega[..., None]  # [N,K,3]
        abar = (w[..., None] * alpha).sum(1, keepdim=True)  # [N,1,3]
        dw = w[..., None] * (abar - alpha)  # [N,K,3]
        J = torch.einsum("nkd,nkc->ndc", v, dw)  # [N,D,3]
        Fdim = feat.shape[-1]
        J[:, Fdim:, :] += w.sum(1)[:, None, None] * torch.eye(3)
        u = dec.sdf_scale * ((act * dec.W2[0]) @ dec.W1)  # [N,D]
        g = torch.einsum("nd,ndc->nc", u, J)
    return sdf, g, nn_counts


# --------------------------------------------------------------------------------------------------
# a8: loss
# --------------------------------------------------------------------------------------------------
def sdf_bce_loss(pred, label, sigma, weight, weighted=False, bce_reduction="mean"):
    """utils/loss.py:44-62."""
    target = torch.sigmoid(label / sigma)
    return F.binary_cross_entropy_with_logits(
        pred / sigma, target, weight=weight if weighted else None, reduction=bce_reduction
    )


def sdf_diff_loss(pred, label, weight, scale=1.0, l2_loss=True):
    """utils/loss.py:9-17 (`main_loss_type` "sdf_l2" / "sdf_l1"; the weight is always applied)."""
    diff_m = (pred - label) / scale
    return (weight * (diff_m**2 if l2_loss else diff_m.abs())).sum() / pred.shape[0]


def sdf_zhong_loss(pred, label, trunc_dist=None, weight=None, weighted=False):
    """utils/loss.py:66-84 (`main_loss_type` "zhong"): L1 distance to the segment between 0 and the label."""
    mid = label / 2.0
    shift_abs = (pred - mid).abs()
    mask = shift_abs > mid.abs()
    loss = torch.where(mask, shift_abs - mid.abs(), torch.zeros_like(label))
    if trunc_dist is not None:
        loss = torch.where(label.abs() < trunc_dist, (pred - label).abs(), loss)
    return (loss * (weight if weighted else 1.0)).mean()


def main_loss(lc, pred, label, weight):
    """The `main_loss_type` switch of utils/mapper.py:751-767."""
    if lc.main_loss_type == "bce":
        return sdf_bce_loss(pred, label, lc.sigma, weight, lc.loss_weight_on)
    if lc.main_loss_type == "zhong":
        return sdf_zhong_loss(pred, label, None, weight, lc.loss_weight_on)
    if lc.main_loss_type in ("sdf_l1", "sdf_l2"):
        return sdf_diff_loss(pred, label, weight, l2_loss=lc.main_loss_type == "sdf_l2")
    raise SystemExit("Please choose a valid loss type")


def transform_batch(points: torch.Tensor, transformation: torch.Tensor) -> torch.Tensor:
    """utils/tools.py:612-636: bmm(R, p) + t with the poses cast to the points' dtype."""
    rot = transformation[:, :3, :3].to(points)
    trans = transformation[:, :3, 3:].to(points)
    return (torch.bmm(rot, points.unsqueeze(-1)) + trans).squeeze(-1)


def eikonal_loss(g: torch.Tensor) -> torch.Tensor:
    """utils/mapper.py:795-797 (`ekional_add_to == "all"`)."""
    return ((g.norm(2, dim=-1) - 1.0) ** 2).mean()


# --------------------------------------------------------------------------------------------------
# a9: Adam exactly as torch.optim.Adam's single-tensor path
# --------------------------------------------------------------------------------------------------
@dataclass
class AdamState:
    m: torch.Tensor
    v: torch.Tensor
    step: int = 0


def adam_step(p: torch.Tensor, g: torch.Tensor, s: AdamState, lr=0.01, b1=0.9, b2=0.99, eps=1e-15, weight_decay=0.0):
    """utils/tools.py:205-255 -> optim.Adam(betas=(0.9,0.99), eps=adam_eps), non-amsgrad."""
    s.step += 1
    if weight_decay != 0.0:
        g = g + weight_decay * p
    s.m.lerp_(g, 1.0 - b1)
    s.v.mul_(b2).addcmul_(g, g, value=1.0 - b2)
    bc1 = 1.0 - b1**s.step
    bc2 = 1.0 - b2**s.step
    denom = (s.v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(s.m, denom, value=-(lr / bc1))


# --------------------------------------------------------------------------------------------------
# a1 + a11: the mapping loop
# --------------------------------------------------------------------------------------------------
@dataclass
class SamplePool:
    """utils/mapper.py:84-97 (the tensors `get_batch` gathers from)."""

    global_coord: torch.Tensor  # [S,3]
    sdf_label: torch.Tensor  # [S]
    time: torch.Tensor  # [S] int32
    weight: torch.Tensor  # [S]
    # Mapper.ba_done_flag (utils/mapper.py:646-658): sensor-frame coordinates + used_poses [frames,4,4]; None = global_coord is current
    local_coord: Optional[torch.Tensor] = None
    used_poses: Optional[torch.Tensor] = None
    # config.proj_correction_on (utils/mapper.py:652-653, 712-714): used_poses [frames,4,4] whose translations are the frames' origins
    frame_poses: Optional[torch.Tensor] = None

    def coord_of(self, index):
        """What the loop body trains on: utils/mapper.py:646-658."""
        if self.used_poses is None:
            return self.global_coord[index]
        return transform_batch(self.local_coord[index], self.used_poses[self.time[index].long()])


@dataclass
class LoopConfig:
    sigma: float = 0.055  # logistic_gaussian_ratio * sigma_sigmoid_m (utils/mapper.py:71)
    loss_weight_on: bool = True
    ekional_loss_on: bool = True
    weight_e: float = 0.5
    numerical_grad: bool = True
    gradient_decimation: int = 10
    fd_eps: float = 0.08  # voxel_size_m * num_grad_step_ratio (utils/mapper.py:703)
    lr: float = 0.01
    adam_eps: float = 1e-15
    weight_decay: float = 0.0
    train_decoder: bool = True  # False after `freeze_model` (utils/tools.py:314, slam.py:193-196)
    loss_scale_counts: Optional[tuple] = None  # (N_global, N'_global) for sharded runs; None = local
    fd_first: int = 0  # sharded runs: local position of the first decimated sample
    ekional_add_to: str = "all"  # utils/mapper.py:779-789: "all" | "surface" | "freespace"
    surface_sample_range_m: float = 0.25  # config: the |sdf_label| threshold of the surface mask (utils/mapper.py:692-694)
    main_loss_type: str = "bce"  # utils/mapper.py:751-767: "bce" | "zhong" | "sdf_l1" | "sdf_l2"
    proj_correction_on: bool = False  # utils/mapper.py:57-69, 712-714: labels scaled by |cos(g, x - origin)|, g = autograd gradient
    # utils/mapper.py:716-741, 770-776: gradient-consistency term between every drawn sample and a randomly shifted copy of it
    consistency_loss_on: bool = False
    weight_c: float = 0.5


def draw_batch_index(pool_count: int, new_idx: Optional[torch.Tensor], bs: int, bs_new_sample: int, gen=None):
    """utils/mapper.py:473-500: bs-bs_new uniform over the pool, bs_new from the new samples."""
    if bs_new_sample > 0 and new_idx is not None and new_idx.shape[0] > 0:
        bs_new = min(new_idx.shape[0], bs_new_sample)
        hist = torch.randint(0, pool_count, (bs - bs_new,), generator=gen)
        pick = torch.randint(0, new_idx.shape[0], (bs_new,), generator=gen)
        return torch.cat((hist, new_idx[pick]), dim=0)
    return torch.randint(0, pool_count, (bs,), generator=gen)


def loss_and_grads(st: MapState, dec: DecoderParams, pool: SamplePool, index: torch.Tensor, lc: LoopConfig, consistency=None):
    """One iteration body of utils/mapper.py:642-835 up to `backward()`; returns a dict with the
    loss triple, sdf_pred and the gradients of the feature table and decoder tensors.
    `consistency` (lc.consistency_loss_on): the iteration's two random draws of utils/mapper.py:717-726 as the caller made them,
    (near_index [n_c] int64, random_shift [N,3] in [-consistency_range, consistency_range))."""
    coord = pool.coord_of(index)
    label = pool.sdf_label[index]
    ts = pool.time[index]
    weight = pool.weight[index].abs()
    theta = st.local_geo_features
    params = [theta] + (dec.tensors() if lc.train_decoder else [])
    for p in params:
        p.requires_grad_(True)
        p.grad = None
    analytic = ((lc.ekional_loss_on and not lc.numerical_grad) or lc.proj_correction_on
                or lc.consistency_loss_on)  # `require_gradient`, utils/mapper.py:57-69
    if analytic:
        coord = coord.clone().requires_grad_(True)  # :660-661
    f, w, _, _, _ = query_feature(st, coord, ts)
    sdf_pred = mlp_sdf(dec, f)
    if not st.weighted_first:
        sdf_pred = (sdf_pred * w).sum(dim=1).squeeze(1)
    g = None
    if analytic:
        g = autograd_gradient(coord, sdf_pred)
    elif lc.ekional_loss_on and lc.numerical_grad:
        g = numerical_gradient(st, dec, coord[lc.fd_first :: lc.gradient_decimation], lc.fd_eps)
    if lc.proj_correction_on:  # utils/mapper.py:652-653, 712-714 (the poses are float64 there: the product promotes)
        origins = pool.frame_poses[ts.long()][:, :3, 3]
        label = label * torch.abs(F.cosine_similarity(g, coord - origins))
    g_near = None
    if lc.consistency_loss_on:  # utils/mapper.py:716-741: the shifted copies, queried with the defaults (training side effects on)
        near_index, random_shift = consistency
        coord_near = (coord + random_shift)[near_index, :]
        f_n, w_n, _, _, _ = query_feature(st, coord_near)
        pred_near = mlp_sdf(dec, f_n)
        if not st.weighted_first:
            pred_near = (pred_near * w_n).sum(dim=1).squeeze(1)
        g_near = autograd_gradient(coord_near, pred_near)
    n_main = sdf_pred.shape[0]
    l_bce = main_loss(lc, sdf_pred, label, weight)  # (named after the default; utils/mapper.py:751-767)
    total = l_bce
    l_cons = torch.zeros(())
    if g_near is not None:  # utils/mapper.py:770-776
        l_cons = (1.0 - F.cosine_similarity(g[near_index, :], g_near)).mean()
        total = total + lc.weight_c * l_cons
    l_eik = torch.zeros(())
    if lc.ekional_loss_on and lc.weight_e > 0 and g is not None:
        g_used = g
        if lc.ekional_add_to != "all":  # utils/mapper.py:779-789 (mask on the decimated subset)
            surface = (pool.sdf_label[index].abs() < lc.surface_sample_range_m)[lc.fd_first :: (lc.gradient_decimation if lc.numerical_grad else 1)]
            g_used = g[surface] if lc.ekional_add_to == "surface" else g[~surface]
        l_eik = eikonal_loss(g_used)
        total = total + lc.weight_e * l_eik
    if lc.loss_scale_counts is not None:  # sharded: normalise by the global counts
        n_glob, ng_glob = lc.loss_scale_counts
        total = l_bce * (n_main / n_glob)
        if g is not None and g.shape[0] > 0:
            total = total + lc.weight_e * l_eik * (g.shape[0] / ng_glob)
    total.backward()
    out = {
        "loss": total.detach(),
        "sdf_loss": l_bce.detach(),
        "eikonal_loss": l_eik.detach(),
        "consistency_loss": l_cons.detach(),
        "sdf_pred": sdf_pred.detach(),
        "g": None if g is None else g.detach(),
        "grad_theta": theta.grad.detach().clone(),
    }
    if lc.train_decoder:
        for name, t in zip(("W1", "b1", "W2", "b2"), dec.tensors()):
            out["grad_" + name] = t.grad.detach().clone()
    for p in params:
        p.requires_grad_(False)
        p.grad = None
    return out


def relu_ambiguous_rows(st: MapState, dec: DecoderParams, pool: SamplePool, index: torch.Tensor, lc: LoopConfig, tau: float,
                        with_slack: bool = False):
    """Checker aid (no reference counterpart): the map rows whose gradient of this iteration is NOT a continuous function of
    the fp32 rounding -- rows gathered by a query point (batch sample or finite-difference copy, utils/mapper.py:697-704)
    that has a hidden pre-activation of the decoder (model/decoder.py:58-82) within `tau` of the ReLU kink.  Two correct
    fp32 evaluations of such a query (different summation orders in W1 f + b1) may open / close that unit, which moves the
    gradient of the query's <= K neighbour rows by one hidden unit's whole contribution.  Returns (rows int64, queries).

    `with_slack`: also a per-row BOUND on that movement, [M + 1] floats (0 for rows no such query gathers): for every
    (query q, unit h) on the kink and every neighbour row j of q
        w_qj * |dL/dsdf_q| * sdf_scale * |W2[h]| * ||W1[h, :F]||_2 * (layer norm: 1 / sqrt(var_j + 1e-5))
    summed per row -- what opening or closing unit h for query q adds to / removes from the gradient of row j (the L2 norm
    of the 8-vector; the layer-norm Jacobian rstd (I - 11^T / F - xh xh^T / F) has norm <= rstd).  dL/dsdf comes from the
    loss of utils/mapper.py:746-798 restated on the SDFs of all bs + 6 n_fd points as leaves.  A checker then holds a listed
    row to strict tolerance + this bound instead of a blanket looser bar.  None where the bound is not derived
    (`weighted_first: False`, analytic eikonal, sharded normalisers).  Fourth value: bool [M + 1], the rows ANY query point of
    the iteration gathers (a row outside it must receive an exactly-zero gradient from any correct implementation).  Fifth
    value: the same bound for the DECODER's gradient, [H D + 2 H + 1] floats in the order W1 | b1 | W2 | b2: opening or closing
    unit h for query q moves dW1[h, c] by |dL/dsdf_q| sdf_scale |W2[h]| |f_q[c]|, db1[h] by the same without f, and dW2[h]
    by at most |dL/dsdf_q| sdf_scale tau (the unit's activation is inside the band); None where not derived."""
    with torch.no_grad():
        coord = pool.coord_of(index)
        pts = [coord]
        n_fd = 0
        if lc.ekional_loss_on and lc.numerical_grad:
            x = coord[lc.fd_first :: lc.gradient_decimation]
            n_fd = x.shape[0]
            for a in range(3):
                e = torch.zeros(3, dtype=x.dtype)
                e[a] = lc.fd_eps
                pts += [x + e, x - e]
        allp = torch.cat(pts, dim=0)
        cert = st.local_point_certainties.clone()
        f, w, _, _, idx = query_feature(st, allp, None, training_mode=False)
        st.local_point_certainties.copy_(cert)
        if not st.weighted_first:
            f = f.reshape(-1, f.shape[-1])
            idx_q = idx.reshape(-1, 1)
        else:
            idx_q = idx
        pre = F.linear(f, dec.W1, dec.b1) if BF16_CONTRACTIONS is None else F.linear(_bf16(f), _bf16(dec.W1)) + dec.b1
        amb_u = pre.abs() < tau
        amb = amb_u.any(dim=1)
        rows = idx_q[amb].reshape(-1)
        rows_u, n_q = torch.unique(rows[rows >= 0]), int(amb.sum())
        gathered = torch.zeros(st.local_geo_features.shape[0], dtype=torch.bool)
        gathered[idx_q[idx_q >= 0]] = True   # rows some query point of the iteration gathers at all
    if not with_slack:
        return rows_u, n_q
    derivable = (st.weighted_first and lc.loss_scale_counts is None and (lc.numerical_grad or not lc.ekional_loss_on))
    if not derivable:
        return rows_u, n_q, None, gathered, None
    bs = coord.shape[0]
    sdf = mlp_sdf(dec, f).detach().requires_grad_(True)
    weight = pool.weight[index].abs()
    total = main_loss(lc, sdf[:bs], pool.sdf_label[index], weight)
    if n_fd > 0 and lc.weight_e > 0:
        s = sdf[bs:].unsqueeze(-1)
        g = torch.cat([(s[2 * a * n_fd:(2 * a + 1) * n_fd] - s[(2 * a + 1) * n_fd:(2 * a + 2) * n_fd]) / (2 * lc.fd_eps)
                       for a in range(3)], dim=1)
        total = total + lc.weight_e * eikonal_loss(g)
    (dsdf,) = torch.autograd.grad(total, sdf)
    with torch.no_grad():
        Fdim = st.local_geo_features.shape[1]
        unit = dec.W2.reshape(-1).abs() * dec.W1[:, :Fdim].norm(dim=1)           # [H]
        per_q = dsdf.abs() * abs(float(dec.sdf_scale)) * (amb_u.to(unit.dtype) * unit).sum(dim=1)   # [Q]
        contrib = (w.reshape(w.shape[0], -1) * per_q[:, None])                     # [Q, K]
        slack = torch.zeros(st.local_geo_features.shape[0], dtype=contrib.dtype)
        ok = (idx_q >= 0) & amb[:, None]
        slack.index_add_(0, idx_q[ok], contrib[ok])
        if st.layer_norm_on:
            var = st.local_geo_features.detach().var(dim=1, unbiased=False)
            slack = slack / torch.sqrt(var + 1e-5)
        dz = dsdf.abs() * abs(float(dec.sdf_scale))                                  # [Q]
        gate = amb_u.to(unit.dtype) * dz[:, None]                                  # [Q, H]
        gw = gate * dec.W2.reshape(-1).abs()[None, :]
        dec_slack = torch.cat(((gw.t() @ f.abs()).reshape(-1), gw.sum(dim=0), gate.sum(dim=0) * tau, torch.zeros(1, dtype=unit.dtype)))
    return rows_u, n_q, slack, gathered, dec_slack


def mapping_iters(st: MapState, dec: DecoderParams, pool: SamplePool, index_seq, lc: LoopConfig, record=False,
                  ambiguity_tau: Optional[float] = None, consistency_seq=None):
    """utils/mapper.py:620-862 with a teacher-forced batch-index sequence (`index_seq` [iters,bs]).

    A NEW Adam state is created per call (utils/mapper.py:634).  Returns the per-iteration records
    when `record`, else the list of loss triples.  `ambiguity_tau`: also record `relu_ambiguous_rows` of every
    iteration (evaluated on the parameters the iteration starts from)."""
    theta = st.local_geo_features
    ad_theta = AdamState(torch.zeros_like(theta), torch.zeros_like(theta))
    ad_dec = [AdamState(torch.zeros_like(t), torch.zeros_like(t)) for t in dec.tensors()]
    recs = []
    for it in range(len(index_seq)):
        amb = relu_ambiguous_rows(st, dec, pool, index_seq[it], lc, ambiguity_tau, with_slack=True) if (record and ambiguity_tau) else None
        out = loss_and_grads(st, dec, pool, index_seq[it], lc, None if consistency_seq is None else consistency_seq[it])
        if amb is not None:
            (out["ambiguous_rows"], out["ambiguous_queries"], out["ambiguous_row_slack"], out["gathered_rows"],
             out["ambiguous_decoder_slack"]) = amb
        with torch.no_grad():
            if lc.train_decoder:
                for name, t, s in zip(("W1", "b1", "W2", "b2"), dec.tensors(), ad_dec):
                    adam_step(t, out["grad_" + name], s, lc.lr, eps=lc.adam_eps)
            adam_step(theta, out["grad_theta"], ad_theta, lc.lr, eps=lc.adam_eps, weight_decay=lc.weight_decay)
        if record:
            out["theta"] = theta.detach().clone()
            out["dec"] = [t.detach().clone() for t in dec.tensors()]
            out["adam_m_theta"] = ad_theta.m.clone()
            out["adam_v_theta"] = ad_theta.v.clone()
            out["certainties"] = st.local_point_certainties.clone()
            out["ts_update"] = st.local_point_ts_update.clone()
            recs.append(out)
        else:
            recs.append((float(out["loss"]), float(out["sdf_loss"]), float(out["eikonal_loss"])))
    return recs


# --------------------------------------------------------------------------------------------------
# a10
# --------------------------------------------------------------------------------------------------
def assign_local_to_global(st: MapState, local_mask: torch.Tensor, point_ts_update: torch.Tensor):
    """model/neural_points.py:538-549 (`local_mask` [Mg+1] bool with the padding slot True)."""
    st.geo_features[local_mask] = st.local_geo_features.detach()
    st.point_certainties[local_mask[:-1]] = st.local_point_certainties
    point_ts_update[local_mask[:-1]] = st.local_point_ts_update


# --------------------------------------------------------------------------------------------------
# N1: tracking measurement model
# --------------------------------------------------------------------------------------------------
def h_model(st: MapState, dec: DecoderParams, rot: torch.Tensor, pos: torch.Tensor, pc_imu: torch.Tensor,
            min_nn: int = 6, min_grad_norm: float = 0.5, max_grad_norm: float = 1.5, max_sdf_std: float = 0.25):
    """utils/error_state_iekf.py:176-264.  Returns (z [Nv] f64, H [Nv,18] f64, valid_points [Nv,3],
    R_inv [Nv] f64, valid mask [N])."""
    T = torch.eye(4)
    T[:3, :3] = rot
    T[:3, 3] = pos
    homo = torch.cat([pc_imu, torch.ones(pc_imu.shape[0], 1)], dim=1)
    pc_map = (homo @ T.T)[:, :3]  # utils/tools.py:590-609
    x = pc_map.clone().requires_grad_(True)
    f, w, nn, cert, _ = query_feature(st, x, training_mode=False)
    sdf = mlp_sdf(dec, f)
    sdf_std = torch.zeros(pc_map.shape[0])
    if not st.weighted_first:
        mean = (sdf * w).sum(dim=1)
        sdf_std = torch.sqrt((w * (sdf - mean.unsqueeze(-1)) ** 2).sum(dim=1)).squeeze(1).detach()
        sdf = mean.squeeze(1)
    g = autograd_gradient(x, sdf).detach()
    sdf = sdf.detach()
    gn = g.norm(dim=-1)
    valid = (nn >= min_nn) & (gn < max_grad_norm) & (gn > min_grad_norm) & (sdf_std < max_sdf_std)
    g, gn, sdf, p = g[valid], gn[valid], sdf[valid], pc_imu[valid]
    n = g.shape[0]
    skew = torch.zeros((n, 3, 3))  # utils/so3_math.py:23-33
    skew[:, 0, 1], skew[:, 0, 2] = -p[:, 2], p[:, 1]
    skew[:, 1, 0], skew[:, 1, 2] = p[:, 2], -p[:, 0]
    skew[:, 2, 0], skew[:, 2, 1] = -p[:, 1], p[:, 0]
    A = rot.to(torch.float32).unsqueeze(0).repeat(n, 1, 1) @ skew
    H = torch.zeros((n, 18), dtype=torch.float64)
    H[:, 0:3] = -(g.unsqueeze(1) @ A).squeeze(1)
    H[:, 3:6] = g
    z = sdf.to(torch.float64)
    ga = (gn - 1.0).to(torch.float64)
    r_inv = 1 / (1 + ga**2) * (0.4 / (0.4 + z**2)) * 1000
    return z, H, pc_map[valid], r_inv, valid
