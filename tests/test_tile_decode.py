"""GPU parity of the matrix-core decode kernels (csrc/train_tile.hip, clid_decode_variant 1 = fp32 MFMA, 2 = bf16 MFMA)
against the 16-lanes-per-query kernel, the reference fixtures (G6) and the CPU oracle.

fp32 variant: same bar as everything else (1e-4, in practice ~1e-6: an MFMA fp32 product chain is an fmaf chain).
bf16 variant (BASELINE.json configs[2]): bf16 operands / fp32 accumulation is NOT inside the 1e-4 bar; the test
prints the measured SDF error and bounds it by what bf16 rounding of an 11-term / 64-term dot product allows."""
import numpy as np
import os

import pytest
import torch

import golden_io as gio
from oracle import cpu_ref as O
import test_hip_parity as T
from test_hip_parity import _fused_grads, maxerr

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import shim_io

    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return shim_io


def _inputs(env, bs, seed=3, ln=False):
    p = gio.load("pool.npz")
    g = gio.load("g6_loop_numerical_train_ln0.npz")
    cfg = env.config(bs=bs, layer_norm_on=ln)
    if bs == int(g["index_seq"].shape[1]):
        index = gio.T(g["index_seq"])[0]
    else:
        gen = torch.Generator().manual_seed(seed)
        index = torch.randint(0, p["coord"].shape[0], (bs,), generator=gen)
    return p, g, cfg, index


@pytest.mark.parametrize("bs,batch_offset,frozen,ln", [(4096, 0, False, 0), (4096, 7, False, 1), (16384, 0, False, 0),
                                                       (16384, 3, True, 0), (16384, 0, False, 1), (1000, 0, False, 0),
                                                       (1000, 5, True, 1), (65536, 0, False, 0)])
def test_tile_fp32_equals_the_valu_decode_kernel(env, bs, batch_offset, frozen, ln):
    """Same search records, the two decode kernels: gradients / losses / certainties equal up to summation order,
    time stamps bit-equal.  Covers ragged task counts (bs = 1000: the last tile is half empty) and the grid-stride
    path (bs = 65536: more tiles than blocks)."""
    p, g, cfg, index = _inputs(env, bs, ln=bool(ln))
    a = _fused_grads(env, cfg, p, g, index, batch_offset=batch_offset, frozen=frozen, split=True, variant=0)
    b = _fused_grads(env, cfg, p, g, index, batch_offset=batch_offset, frozen=frozen, split=True, variant=1)
    scale = max(1.0, float(a[0].abs().max()))
    assert maxerr(a[0], b[0]) <= 5e-6 * scale
    if frozen:
        assert not b[0][:833].any()
    assert maxerr(a[1], b[1]) <= 2e-6
    assert maxerr(a[2], b[2]) <= 1e-5 * max(1.0, float(a[2].abs().max()))  # certainties grow with the batch size
    assert torch.equal(a[3], b[3])
    # exact zeros stay exact: rows no query touched
    ga, gb = a[0][836:].view(-1, 8), b[0][836:].view(-1, 8)
    assert torch.equal((ga == 0).all(1), (gb == 0).all(1))


def test_tile_fp32_gradients_vs_reference(env):
    """One iteration of the tile kernel against the REFERENCE's own gradients (G6 it0)."""
    from clid_slam_amd import _lib

    bs = int(gio.load("g6_loop_numerical_train_ln0.npz")["index_seq"].shape[1])
    p, g, cfg, index = _inputs(env, bs)
    grad, loss, cert, ts = _fused_grads(env, cfg, p, g, index, split=True, variant=1)
    assert abs(float(loss[1]) - float(g["loss_bce"][0])) <= 5e-6
    assert abs(float(loss[0]) - float(g["loss_total"][0])) <= 5e-6
    H, D = _lib.H, _lib.D
    parts = {"W1": grad[: H * D].view(H, D), "b1": grad[H * D: H * D + H],
             "W2": grad[H * D + H: H * D + 2 * H].view(1, H), "b2": grad[H * D + 2 * H: H * D + 2 * H + 1]}
    for n, t in parts.items():
        ref = g[f"it0_grad_{n}"]
        assert np.abs(t.numpy() - ref).max() <= 1e-4 * max(np.abs(ref).max(), 1e-12), n
    gt = grad[_lib.GRAD_FEAT_OFFSET:].view(-1, 8).numpy()
    rows = g["it0_grad_theta_rows"].astype(np.int64)
    dense = np.zeros_like(gt)
    dense[rows] = g["it0_grad_theta_vals"]
    assert np.abs(gt - dense).max() <= 1e-4 * np.abs(dense).max()
    untouched = np.ones(gt.shape[0], bool)
    untouched[rows] = False
    assert not gt[untouched].any()
    assert maxerr(cert, g["it0_certainties"]) <= 1e-5 * max(1.0, float(np.abs(g["it0_certainties"]).max()))
    assert np.array_equal(ts.numpy(), g["it0_ts_update"])


def _oracle_sdf(rec, g, ln=False):
    """SDF of the oracle at the query position of every live record slot."""
    qi = rec[:, 0:8, :]  # [tasks, 8, (x, y, z, stamp bits)]
    live = qi[..., 3].contiguous().view(torch.int32) >= 0
    x = qi[..., :3][live].contiguous()
    st = gio.map_state()
    st.layer_norm_on = bool(ln)
    st.local_geo_features = gio.T(gio.load("pool.npz")["base_geo_features"])[gio.T(gio.load("state.npz")["local_mask"])].clone()
    dec = gio.decoder(g, "init_")
    with torch.no_grad():
        s = O.sdf_at(st, dec, x)
    return live, s


@pytest.mark.parametrize("variant", [1, 2])
def test_tile_sdf_vs_oracle(env, variant):
    """max |SDF_hip - SDF_oracle| over every query point of one iteration (batch + finite-difference copies)."""
    p, g, cfg, index = _inputs(env, 4096)
    out = []
    _fused_grads(env, cfg, p, g, index, split=True, variant=variant, sdf_out=out)
    rec, sdf = out[0]
    live, ref = _oracle_sdf(rec, g)
    err = float((sdf[live] - ref).abs().max())
    rng = float(ref.abs().max())
    print(f"\n[tile decode variant {variant}] max|dSDF| vs oracle = {err:.3e} over {int(live.sum())} query points "
          f"(|SDF| up to {rng:.3f} m)")
    if variant == 1:
        assert err <= 1e-5          # north-star bar is 1e-4
    else:
        assert 1e-7 < err <= 3e-3   # bf16 operands: ~2^-9 relative per product, sdf_scale 0.055


def test_mapping_loop_g6_on_the_valu_kernel(env):
    """The tile kernel is the default decode kernel; the 16-lane kernel must stay green on the reference loop too."""
    from clid_slam_amd import _lib

    prev, _lib.DECODE_VARIANT = _lib.DECODE_VARIANT, 0  # the process default (CLID_DECODE) of Mappers without their own
    try:
        T.test_mapping_loop_g6(env, "numerical", False, 0)
        T.test_mapping_loop_g6(env, "numerical", False, 1)
        T.test_mapping_loop_g6(env, "numerical", True, 0)
    finally:
        _lib.DECODE_VARIANT = prev


def test_default_decode_kernel_is_the_tile_kernel(env):
    import ctypes as C

    from clid_slam_amd import _lib

    lib = _lib.load()
    p, g, cfg, _ = _inputs(env, 4096)
    nm = env.neural_points(cfg, base=p)
    view, keep = nm._map_view(True)
    ta = _lib.TrainArgs()
    assert _lib.DECODE_VARIANT == int(os.environ.get("CLID_DECODE", "1"))
    ta.eikonal_mode, ta.grad_stride, ta.decode_variant = 1, _lib.GRAD_ROW16, _lib.DECODE_VARIANT
    assert lib.clid_train_decode_kernel(C.byref(view), C.byref(ta)) == _lib.DECODE_VARIANT
    ta.eikonal_mode = 2  # analytic eikonal: 16-lane kernel family
    assert lib.clid_train_decode_kernel(C.byref(view), C.byref(ta)) == 0


def test_bf16_mapping_tracks_fp32(env):
    """BASELINE.json configs[2] (bf16 decoder contractions, fp32 master weights / accumulation): 10 iterations from
    the same state as the fp32 run.  The loss trajectory must track the fp32 one closely; parameters are compared
    statistically (Adam with eps = 1e-15 turns sign differences of tiny gradients into +-lr steps)."""
    from clid_slam_amd import _lib

    lib = _lib.load()
    p, g, cfg, _ = _inputs(env, 16384)
    gen = torch.Generator().manual_seed(5)
    idx = torch.randint(0, p["coord"].shape[0], (10, 16384), generator=gen).cuda()
    res = {}
    for variant in (1, 2):
        nm = env.neural_points(cfg, base=p)
        dec = env.decoder(cfg, g, "init_")
        mp, _ = env.mapper(cfg, nm, dec)
        mp.decode_variant = variant
        mp.mapping(10, index_seq=idx)
        torch.cuda.synchronize()
        res[variant] = (mp.last_losses.cpu(), nm.local_geo_features.detach().cpu().clone(),
                        [t.detach().cpu().clone() for t in dec.flat_params()])
    l32, l16 = res[1][0], res[2][0]
    dl = float((l32[:, 0] - l16[:, 0]).abs().max())
    dtheta = (res[1][1] - res[2][1]).abs()
    print(f"\n[bf16 vs fp32, 10 iterations bs 16384] max|dloss| {dl:.2e} (loss {float(l32[-1, 0]):.4f}); "
          f"theta: mean|d| {float(dtheta.mean()):.2e}, max|d| {float(dtheta.max()):.2e}")
    assert dl <= 5e-3 * max(1.0, float(l32[:, 0].abs().max()))
    assert float(l16[-1, 0]) < float(l16[0, 0])  # it trains
    assert float(dtheta.mean()) <= 2e-3


@pytest.mark.parametrize("variant,ln", [(2, False), (2, True), (1, False), (1, True)])
def test_tile_sdf_vs_oracle_at_65536_samples(env, variant, ln):
    """BASELINE.json configs[2] at its own size: 65 536 samples per iteration put the launch beyond 2048 tiles, where the
    launcher picks the 2-waves-per-block, grid-stride instantiation `k_decode_tile<PREC, LN, 2>` -- a different code object
    from the one the 4096-sample tests run.  SDF of every query point (batch + finite-difference copies, ~105 k) against
    the CPU oracle, the loss against the oracle's loss on the same batch; bf16 operands report their error and are bounded
    by what bf16 rounding of the 11- and 64-term dot products allows, fp32 keeps the 1e-5 bar."""
    bs = 65536
    p, g, cfg, index = _inputs(env, bs, seed=9, ln=ln)
    out = []
    grad, loss, cert, ts = _fused_grads(env, cfg, p, g, index, split=True, variant=variant, sdf_out=out)
    rec, sdf = out[0]
    assert rec.shape[0] > 2 * 2048  # tasks: > 2048 tiles, i.e. the large-launch instantiation ran
    live, ref = _oracle_sdf(rec, g, ln)
    got = sdf[live]
    err = float((got - ref).abs().max())
    # the oracle's loss on the same batch (BCE over the samples + the numerical eikonal term on every 10th)
    st = gio.map_state()
    st.layer_norm_on = bool(ln)
    st.local_geo_features = gio.T(gio.load("pool.npz")["base_geo_features"])[gio.T(gio.load("state.npz")["local_mask"])].clone()
    pool, _ = gio.sample_pool()
    recs = O.mapping_iters(st, gio.decoder(g, "init_"), pool, index[None].to(torch.int64), O.LoopConfig(), record=True)
    dl = abs(float(loss[0]) - float(recs[0]["loss"]))
    print(f"\n[tile decode variant {variant}, layer norm {ln}, bs {bs}] max|dSDF| vs oracle = {err:.3e} over {int(live.sum())} "
          f"query points (|SDF| up to {float(ref.abs().max()):.3f} m); loss {float(loss[0]):.6f} vs oracle {float(recs[0]['loss']):.6f}")
    if variant == 1:
        assert err <= 1e-5 and dl <= 5e-6
    else:
        # layer norm feeds unit-variance features (|f| up to ~2.5 instead of ~0.5): the bf16 product error scales with them
        assert 1e-7 < err <= (1.5e-2 if ln else 3e-3) and dl <= 5e-3


@pytest.mark.parametrize("variant,ln", [(1, False), (1, True), (2, False)])
def test_tile_vs_oracle_at_262144_samples(env, variant, ln):
    """BASELINE.json configs[3] at its own size (262 144 samples per iteration, ~420 k query points, 26 k tiles on the
    large-launch instantiation): the SDF of EVERY query point, the loss, and -- fp32 -- every gradient entry of the feature
    table and the decoder against the CPU oracle's autograd on the same batch (utils/mapper.py:642-836,
    model/decoder.py:58-82).  Rows the oracle names as gathered by a query on the ReLU kink are held to the strict bar + their
    own bound (oracle.cpu_ref.relu_ambiguous_rows); rows nobody gathers are exactly zero."""
    from clid_slam_amd import _lib

    bs = 262144 if variant == 1 else 131072  # (the bf16 kernel is outside the 1e-4 bar at any size: half the oracle time)
    p, g, cfg, index = _inputs(env, bs, seed=11, ln=ln)
    out = []
    grad, loss, cert, ts = _fused_grads(env, cfg, p, g, index, split=True, variant=variant, sdf_out=out)
    rec, sdf = out[0]
    assert rec.shape[0] > 8 * 2048
    live, ref = _oracle_sdf(rec, g, ln)
    err = float((sdf[live] - ref).abs().max())
    st = gio.map_state()
    st.layer_norm_on = bool(ln)
    st.local_geo_features = gio.T(gio.load("pool.npz")["base_geo_features"])[gio.T(gio.load("state.npz")["local_mask"])].clone()
    pool, _ = gio.sample_pool()
    dec = gio.decoder(g, "init_")
    idx64 = index.to(torch.int64)
    lc = O.LoopConfig()
    rows, nq, slack, gathered, dec_slack = O.relu_ambiguous_rows(st, dec, pool, idx64, lc, 4e-6, with_slack=True)
    o = O.loss_and_grads(st, dec, pool, idx64, lc)
    dl = abs(float(loss[0]) - float(o["loss"]))
    gt = grad[_lib.GRAD_FEAT_OFFSET:].view(-1, 8)
    g0 = o["grad_theta"]
    gmax = float(g0.abs().max())
    d = (gt - g0).abs().max(1).values
    excess = float((d - 1.25 * slack).max()) / gmax
    H, D = _lib.H, _lib.D
    # decoder gradients: sums over ~420 k query points.  The fp32 autograd of the CPU oracle forms dW2 = dsdf^T h as one long
    # fp32 dot product per hidden unit, whose accuracy depends on the host BLAS (2e-4 of the largest entry was seen on one
    # box, 2e-6 on another); the SAME oracle in float64 is the arbiter here
    st64 = gio.as_double(gio.map_state())
    st64.layer_norm_on = bool(ln)
    st64.local_geo_features = st.local_geo_features.detach().double()
    o64 = O.loss_and_grads(st64, gio.as_double(gio.decoder(g, "init_")), gio.as_double(gio.sample_pool()[0]), idx64, lc)
    gd = torch.cat([o64["grad_" + n].reshape(-1) for n in ("W1", "b1", "W2", "b2")])
    # (entries of hidden units that some query holds on the ReLU kink may move by the oracle's own bound, like the listed rows)
    ddec = float(((grad[: H * D + 2 * H + 1].double() - gd).abs() - 1.25 * dec_slack.double()).max()) / float(gd.abs().max())
    gd32 = torch.cat([o["grad_" + n].reshape(-1) for n in ("W1", "b1", "W2", "b2")])
    print(f"[decoder gradient, largest entry {float(gd.abs().max()):.3e}] HIP vs float64 oracle {ddec:.2e}; fp32 oracle vs float64 "
          f"oracle {float((gd32.double() - gd).abs().max()) / float(gd.abs().max()):.2e}")
    print(f"\n[tile decode variant {variant}, layer norm {ln}, bs {bs}] max|dSDF| = {err:.3e} over {int(live.sum())} query points; "
          f"loss {float(loss[0]):.6f} vs oracle {float(o['loss']):.6f}; grad theta rel {float(d.max()) / gmax:.2e} "
          f"(beyond the kink bound {excess:.2e}, {len(rows)} listed rows of {nq} queries); decoder grad rel {ddec:.2e}")
    assert not bool((gt != 0).any(1)[~gathered].any())
    if variant == 1:
        assert err <= 1e-5 and dl <= 5e-6
        assert excess <= 1e-4 and ddec <= 1e-4
    else:
        assert 1e-7 < err <= 3e-3 and dl <= 5e-3


@pytest.mark.parametrize("bs,ln", [(65536, False), (65536, True), (4096, False)])
def test_bf16_tile_against_the_oracle_with_bf16_contractions(env, bs, ln, monkeypatch):
    """BASELINE.json configs[2] against the VALUES its arithmetic defines, not against the fp32 oracle at a widened bar: the
    oracle's decoder layer 1 is switched to the bf16 instantiation's contractions (oracle.cpu_ref.BF16_CONTRACTIONS: operands
    rounded to nearest-even bf16, fp32 products and sums; forward pre-activations, d f = bf16(dh) bf16(W1),
    dW1 = bf16(dh)^T bf16(f), db1 = sum bf16(dh)), everything else fp32 as in the kernel.  What is left between the two sides is fp32 summation order plus operands that sit within an ulp of a bf16 rounding
    boundary (a handful per launch, each worth 2^-8 of ONE product): SDF of every query point, the loss, every entry of the
    feature-table gradient (rows on the ReLU kink at the strict bar + the oracle's own per-row bound, as in the fp32 tests; rows
    nobody gathers exactly zero) and of the decoder gradient (the float64 run of the same bf16 arithmetic arbitrates the long
    sums).  Also printed: the distance of both to the fp32 oracle -- what bf16 costs (outside north_star's 1e-4, SURVEY hard
    part 5)."""
    from clid_slam_amd import _lib

    p, g, cfg, index = _inputs(env, bs, seed=9, ln=ln)
    out = []
    grad, loss, cert, ts = _fused_grads(env, cfg, p, g, index, split=True, variant=2, sdf_out=out)
    rec, sdf = out[0]
    assert (rec.shape[0] > 2 * 2048) == (bs >= 65536)  # tasks: 65 536 samples run the launch of several tiles per wave
    live, ref32 = _oracle_sdf(rec, g, ln)
    # (the bf16 instantiations contract dW1 / db1 per wave on the bf16 MFMA at every launch size: the fp32 block-level
    # contraction of the one-tile-per-wave launches is an fp32-only instantiation, csrc/train_tile.hip BLK)
    monkeypatch.setattr(O, "BF16_CONTRACTIONS", {"dW1": True})
    _, ref16 = _oracle_sdf(rec, g, ln)
    err16, err32 = float((sdf[live] - ref16).abs().max()), float((sdf[live] - ref32).abs().max())

    def state(double=False):
        st = gio.map_state()
        st.layer_norm_on = bool(ln)
        st.local_geo_features = gio.T(gio.load("pool.npz")["base_geo_features"])[gio.T(gio.load("state.npz")["local_mask"])].clone()
        return gio.as_double(st) if double else st

    idx64 = index.to(torch.int64)
    lc = O.LoopConfig()
    st, dec, pool = state(), gio.decoder(g, "init_"), gio.sample_pool()[0]
    rows, nq, slack, gathered, dec_slack = O.relu_ambiguous_rows(st, dec, pool, idx64, lc, 4e-6, with_slack=True)
    o = O.loss_and_grads(st, dec, pool, idx64, lc)
    dl = abs(float(loss[0]) - float(o["loss"]))
    gt = grad[_lib.GRAD_FEAT_OFFSET:].view(-1, 8)
    g0 = o["grad_theta"]
    gmax = float(g0.abs().max())
    d = (gt - g0).abs().max(1).values
    excess = float((d - 1.25 * slack).max()) / gmax
    H, D = _lib.H, _lib.D
    # (float64 accumulation of the SAME rounded operands: _bf16 rounds through bfloat16 whatever the carrier type)
    o64 = O.loss_and_grads(state(True), gio.as_double(gio.decoder(g, "init_")), gio.as_double(gio.sample_pool()[0]), idx64, lc)
    gd = torch.cat([o64["grad_" + n].reshape(-1) for n in ("W1", "b1", "W2", "b2")])
    ddec = float(((grad[: H * D + 2 * H + 1].double() - gd).abs() - 1.25 * dec_slack.double()).max()) / float(gd.abs().max())
    monkeypatch.setattr(O, "BF16_CONTRACTIONS", None)
    o32 = O.loss_and_grads(state(), gio.decoder(g, "init_"), gio.sample_pool()[0], idx64, lc)
    cost_theta = float((g0 - o32["grad_theta"]).abs().max()) / gmax
    print(f"\n[bf16 tile decode, bs {bs}, layer norm {ln}] vs the oracle WITH bf16 contractions: max|dSDF| {err16:.2e} over {int(live.sum())} "
          f"query points, dloss {dl:.2e}, grad theta rel {float(d.max()) / gmax:.2e} (beyond the kink bound {excess:.2e}; {len(rows)} rows of "
          f"{nq} queries listed), decoder grad rel {ddec:.2e}  ||  what bf16 costs against the fp32 oracle: max|dSDF| {err32:.2e}, "
          f"grad theta rel {cost_theta:.2e}, dloss {abs(float(o['loss']) - float(o32['loss'])):.2e}")
    assert not bool((gt != 0).any(1)[~gathered].any())
    assert err32 > 1e-7  # (the bf16 kernel really ran)
    # an operand within an fp32 ulp of a bf16 rounding boundary rounds the other way on one side: 2^-8 of one of the 11 products of
    # one pre-activation, i.e. <= sdf_scale |W2_h| 2^-8 |W1_hc f_c| on that query's SDF -- a few 1e-5 at these magnitudes
    assert err16 <= 1e-4 and dl <= 5e-6
    assert excess <= 2e-4 and ddec <= 2e-4
