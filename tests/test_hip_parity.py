"""GPU parity: the HIP path (through the C ABI and the drop-in classes) against the CPU oracle and
the reference-generated golden fixtures, same inputs.  Tolerances: indices / counts / stamps
bit-exact; fp32 values within 1e-4 (north_star), most far tighter."""
import numpy as np
import pytest
import torch

import golden_io as gio
from oracle import cpu_ref as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import shim_io

    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return shim_io


def maxerr(a, b):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    return float(np.abs(a.astype(np.float64) - b.astype(np.float64)).max()) if a.size else 0.0


def cert_close(got, want, rel=1e-5):
    """Certainties are sums of IDW weights over every query pass that touched the row (model/neural_points.py:714): they grow
    with batch size x iterations, and two fp32 summation orders differ by a few ulp of the SUM.  Compared at 1e-5 of the
    largest certainty (>= 1), the form tests/test_tile_decode.py uses -- not at an absolute figure chosen per test."""
    want_t = torch.as_tensor(want).detach().cpu().double()
    return maxerr(got, want) <= rel * max(1.0, float(want_t.abs().max()))


@pytest.mark.parametrize("tf", [False, True])
def test_radius_search_g1(env, tf):
    g = gio.load("g1_search.npz")
    cfg = env.config()
    nm = env.neural_points(cfg)
    d2, idx = nm.radius_neighborhood_search(gio.T(g["x"]).cuda(), time_filtering=tf)
    assert idx.dtype == torch.int64
    assert np.array_equal(idx.cpu().numpy().astype(np.int32), g[f"idx_tf{int(tf)}"])
    ref = g[f"dist2_tf{int(tf)}"]
    got = d2.cpu().numpy()
    if tf:
        # time-filtered foreign ids are dropped at table-build time: their dist2 is the sentinel
        # max_valid_dist2, exactly what the reference writes for idx == -1 (np.py:1013)
        assert np.array_equal(got, ref)
    else:
        assert np.array_equal(got, ref)


CASES = [(ln, wf, tm, loc) for ln in (0, 1) for wf in (1, 0) for tm in (1, 0) for loc in (1, 0) if loc or not (tm or ln)]


@pytest.mark.parametrize("ln,wf,tm,loc", CASES)
def test_query_feature_g2(env, ln, wf, tm, loc):
    g = gio.load("g2_query.npz")
    cfg = env.config(layer_norm_on=bool(ln), weighted_first=bool(wf))
    nm = env.neural_points(cfg)
    tag = f"ln{ln}_wf{wf}_tm{tm}_loc{loc}"
    x, ts = gio.T(g["x"]).cuda(), gio.T(g["ts"]).cuda()
    f, col, w, nn, cert = nm.query_feature(x, ts if loc else None, training_mode=bool(tm), query_locally=bool(loc))
    assert col is None and nn.dtype == torch.int64 and w.shape == (x.shape[0], 6, 1)
    assert np.array_equal(nn.cpu().numpy().astype(np.int32), g["nn_" + tag])
    assert maxerr(f, g["f_" + tag]) <= 2e-6
    assert maxerr(w, g["w_" + tag]) <= 1e-6
    assert maxerr(cert, g["cert_" + tag]) <= 1e-5
    if loc:
        assert maxerr(nm.local_point_certainties, g["post_cert_" + tag]) <= 1e-4
        assert np.array_equal(nm.local_point_ts_update.cpu().numpy(), g["post_ts_" + tag])


def test_mlp_g3(env):
    g = gio.load("g3_mlp.npz")
    cfg = env.config()
    dec = env.decoder(cfg)
    s = dec.sdf(gio.T(g["f"]).cuda())
    assert s.shape == (g["f"].shape[0],)
    assert maxerr(s, g["sdf"]) <= 1e-6
    m = dec.mlp(gio.T(g["f"]).cuda())
    assert m.shape == (g["f"].shape[0], 1)
    assert maxerr(m.squeeze(1) * dec.sdf_scale, g["sdf"]) <= 1e-6


@pytest.mark.parametrize("ln", [0, 1])
def test_analytic_gradient_g4(env, ln):
    from clid_slam_amd.tools import get_gradient

    g = gio.load("g4_grad.npz")
    cfg = env.config(layer_norm_on=bool(ln))
    nm = env.neural_points(cfg)
    dec = env.decoder(cfg)
    x = gio.T(g["x"]).cuda()
    # fused inference kernel
    s, gr, nn, _ = nm.query_sdf_and_gradient(dec, x)
    assert maxerr(s, g[f"sdf_ln{ln}"]) <= 2e-6
    assert maxerr(gr, g[f"grad_ln{ln}"]) <= 5e-5
    assert np.array_equal(nn.cpu().numpy().astype(np.int32), g[f"nn_ln{ln}"])
    # the reference's own call sequence through autograd (error_state_iekf.py:209-227)
    xg = x.clone().requires_grad_(True)
    f, _, w, nn2, _ = nm.query_feature(xg, training_mode=False)
    s2 = dec.sdf(f)
    g2 = get_gradient(xg, s2)
    assert maxerr(s2, g[f"sdf_ln{ln}"]) <= 2e-6
    assert maxerr(g2, g[f"grad_ln{ln}"]) <= 5e-5


def test_loss_g5(env):
    from clid_slam_amd.loss import sdf_bce_loss

    g = gio.load("g5_loss.npz")
    pred = gio.T(g["pred"]).cuda().requires_grad_(True)
    l = sdf_bce_loss(pred, gio.T(g["label"]).cuda(), 0.055, gio.T(g["weight"]).cuda(), True)
    l.backward()
    assert abs(float(l.detach()) - float(g["l_bce"])) <= 2e-6
    assert maxerr(pred.grad, g["dpred"]) <= 1e-8


def test_autograd_backward_vs_oracle(env):
    """d/d(theta, x, decoder) of a scalar of the un-fused sequence == oracle autograd."""
    g = gio.load("g2_query.npz")
    for ln, wf in ((0, 1), (1, 1), (0, 0), (1, 0)):
        cfg = env.config(layer_norm_on=bool(ln), weighted_first=bool(wf))
        nm = env.neural_points(cfg)
        dec = env.decoder(cfg)
        x = gio.T(g["x"])[:512]
        torch.manual_seed(0)
        # oracle
        st = gio.map_state(layer_norm_on=bool(ln), weighted_first=bool(wf))
        dp = gio.decoder()
        st.local_geo_features.requires_grad_(True)
        for t in dp.tensors():
            t.requires_grad_(True)
        xo = x.clone().requires_grad_(True)
        fo, wo, _, _, _ = O.query_feature(st, xo, training_mode=False)
        if wf:
            so = O.mlp_sdf(dp, fo)
        else:
            h = torch.relu(torch.nn.functional.linear(fo, dp.W1, dp.b1))
            so = ((torch.nn.functional.linear(h, dp.W2, dp.b2) * dp.sdf_scale) * wo).sum(1).squeeze(1)
        coef = torch.linspace(-1, 1, so.shape[0])
        (so * coef).sum().backward()
        # HIP
        xh = x.cuda().requires_grad_(True)
        fh, _, wh, _, _ = nm.query_feature(xh, training_mode=False)
        sh = dec.sdf(fh)
        if not wf:
            sh = (sh * wh).sum(1).squeeze(1)
        (sh * coef.cuda()).sum().backward()
        assert maxerr(sh, so) <= 2e-6
        assert maxerr(xh.grad, xo.grad) <= 5e-5, (ln, wf)
        gth = nm.local_geo_features.grad
        gto = st.local_geo_features.grad
        assert maxerr(gth, gto) <= 1e-5 * max(1.0, float(gto.abs().max())), (ln, wf)
        assert not gth.cpu()[gto.abs().sum(1) == 0].any()
        for ph, po in zip(dec.flat_params(), dp.tensors()):
            assert maxerr(ph.grad, po.grad) <= 1e-5 * max(1.0, float(po.grad.abs().max()))


def test_adam_kernel_vs_torch(env):
    from clid_slam_amd import _lib

    lib = _lib.load()
    torch.manual_seed(1)
    p0 = torch.randn(1000, 8)
    pr = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([pr], lr=0.01, betas=(0.9, 0.99), eps=1e-15)
    p = p0.clone().cuda()
    m = torch.zeros_like(p)
    v = torch.zeros_like(p)
    for step in range(1, 5):
        g = torch.randn(1000, 8)
        g[::3] = 0.0
        pr.grad = g.clone()
        opt.step()
        gd = g.cuda()
        _lib.check(lib.clid_adam_step(p.data_ptr(), gd.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), 0.01, 0.9,
                                      0.99, 1e-15, 0.0, step, 1, _lib.stream()), "adam")
        assert not gd.any()  # zeroed in the same pass
        assert maxerr(p, pr) <= 5e-7  # <= 1 ulp at |p| < 4 (ATen may fuse the lerp)
    # untouched rows (always-zero gradient) are bit-identical to the start
    assert torch.equal(p.cpu()[::3] , p0[::3]) or maxerr(p.cpu()[::3], p0[::3]) == 0.0


G6 = [("numerical", False, 0), ("numerical", False, 1), ("numerical", True, 0), ("analytic", False, 0), ("analytic", True, 0)]


@pytest.mark.parametrize("add_to", ["surface", "freespace"])
def test_mapping_loop_g6_ekional_add_to(env, add_to):
    """config.ekional_add_to "surface" / "freespace" (utils/mapper.py:779-789) against the reference's own loop: the eikonal
    mean runs over the decimated samples with |sdf_label| below / not below config.surface_sample_range_m."""
    g_all = gio.load("g6_loop_numerical_train_ln0.npz")
    g = gio.load(f"g6_loop_numerical_train_ln0_eik{add_to}.npz")
    assert abs(float(g["loss_total"][0]) - float(g_all["loss_total"][0])) > 2e-5  # (the fixture really differs from "all")
    test_mapping_loop_g6(env, "numerical", False, 0, add_to)


@pytest.mark.parametrize("loss_type", ["sdf_l1", "sdf_l2", "zhong"])
def test_mapping_loop_g6_main_loss_type(env, loss_type):
    """config.main_loss_type "sdf_l1" / "sdf_l2" / "zhong" (utils/mapper.py:751-767, utils/loss.py:9-17, 66-84) against the
    reference's own loop, on the tile decode kernels."""
    g_bce = gio.load("g6_loop_numerical_train_ln0.npz")
    g = gio.load(f"g6_loop_numerical_train_ln0_{loss_type}.npz")
    assert abs(float(g["loss_total"][0]) - float(g_bce["loss_total"][0])) > 1e-3  # (the fixture really is another loss)
    test_mapping_loop_g6(env, "numerical", False, 0, "all", loss_type)


def test_mapping_loop_g6_ba_done_flag(env):
    """Mapper.ba_done_flag (utils/mapper.py:646-658): the batch comes from the sensor-frame pool and every sample is moved by the
    pose of its own frame (utils/tools.py:612-636) -- inside the search launch's gather here; the (stale) world-frame pool is not
    read by the loop."""
    test_mapping_loop_g6(env, "numerical", False, 0, "all", "bce", True)


@pytest.mark.parametrize("ln", [0, 1])
def test_mapping_loop_g6_weighted_first_false_analytic(env, ln):
    """`neuralpoints.weighted_first: False` with `loss.numerical_grad_on: False` (utils/mapper.py:679-680, 695-696: every
    neighbour decoded, the SDFs blended, the eikonal term on the autograd gradient of that blend) against the reference's own
    loop and its double backward: k_train_analytic_wf0 (csrc/train_analytic.hip)."""
    test_mapping_loop_g6(env, "analytic", False, ln, wf=False)


@pytest.mark.parametrize("ln", [0, 1])
def test_mapping_loop_g6_proj_correction(env, ln):
    """config.proj_correction_on (utils/mapper.py:57-69, 695-696, 712-714): every label scaled by |cos(g, x - origin of the sample's
    frame)| with the autograd gradient g of the sample in the graph -- `require_gradient` overrides `numerical_grad`, the eikonal term
    runs on that g over the whole batch, and the BCE term back-propagates through g too.  Against the reference's own loop."""
    g0 = gio.load(f"g6_loop_analytic_train_ln{ln}.npz") if ln == 0 else None
    g = gio.load(f"g6_loop_numerical_train_ln{ln}_proj.npz")
    if g0 is not None:
        assert abs(float(g["loss_bce"][0]) - float(g0["loss_bce"][0])) > 1e-3  # (the correction really changes the targets)
    test_mapping_loop_g6(env, "numerical", False, ln, proj=True)


@pytest.mark.parametrize("ln", [0, 1])
def test_mapping_loop_g6_consistency_loss(env, ln):
    """config.consistency_loss_on (utils/mapper.py:716-741, 770-776): 1 - cos between the autograd gradient of drawn samples and of
    randomly shifted copies of them (a second query with the training side effects on), weighted by weight_c, back-propagated
    through both gradients -- against the reference's own loop with its recorded draws (near_index, random_shift)."""
    g0 = gio.load(f"g6_loop_analytic_train_ln{ln}.npz") if ln == 0 else None
    g = gio.load(f"g6_loop_numerical_train_ln{ln}_cons.npz")
    if g0 is not None:
        assert float(g["loss_total"][0]) - float(g0["loss_total"][0]) > 1e-3  # (the term is really there)
    test_mapping_loop_g6(env, "numerical", False, ln, cons=True)


def test_consistency_loss_with_its_own_draws_runs(env):
    """config.consistency_loss_on without teacher-forced draws (near_index / random_shift from torch's generator, as in the
    reference): the loop runs, every loss row carries the term (column 3 = mean 1 - cos in (0, 2)), the total is bce + weight_e eik
    + weight_c term, and training reduces the total."""
    g = gio.load("g6_loop_numerical_train_ln0.npz")
    p = gio.load("pool.npz")
    cfg = env.config(bs=2048, bs_new_sample=200, consistency_loss_on=True)
    cfg.consistency_count = 512
    nm = env.neural_points(cfg, base=p)
    dec = env.decoder(cfg, g, "init_")
    mp, _ = env.mapper(cfg, nm, dec)
    torch.manual_seed(3)
    mp.mapping(6)
    L = mp.last_losses.cpu().numpy()
    assert L.shape == (6, 4) and np.isfinite(L).all()
    assert (L[:, 3] > 0).all() and (L[:, 3] < 2).all()
    assert np.abs(L[:, 0] - (L[:, 1] + cfg.weight_e * L[:, 2] + cfg.weight_c * L[:, 3])).max() <= 1e-5
    assert L[-1, 0] < L[0, 0]


@pytest.mark.parametrize("mode,frozen,ln", G6)
def test_mapping_loop_g6(env, mode, frozen, ln, add_to="all", loss_type="bce", ba=False, wf=True, proj=False, cons=False):
    from clid_slam_amd.tools import freeze_model

    tag = (f"{mode}_{'frozen' if frozen else 'train'}_ln{ln}" + ("" if add_to == "all" else f"_eik{add_to}")
           + ("" if loss_type == "bce" else f"_{loss_type}") + ("_ba" if ba else "") + ("" if wf else "_wf0") + ("_proj" if proj else "")
           + ("_cons" if cons else ""))
    g = gio.load(f"g6_loop_{tag}.npz")
    p = gio.load("pool.npz")
    cfg = env.config(layer_norm_on=bool(ln), bs=int(g["index_seq"].shape[1]), bs_new_sample=200, ekional_add_to=add_to,
                     main_loss_type=loss_type, weighted_first=bool(wf), proj_correction_on=bool(proj), consistency_loss_on=bool(cons))
    if cons:
        cfg.consistency_count, cfg.weight_c = int(g["cons_near_index"].shape[1]), float(g["cons_weight_c"])
    if mode == "analytic":
        cfg.numerical_grad, cfg.gradient_decimation = False, 1
    nm = env.neural_points(cfg, base=p)
    dec = env.decoder(cfg, g, "init_")
    if frozen:
        freeze_model(dec)
    mp, _ = env.mapper(cfg, nm, dec)
    if proj:
        mp.used_poses = gio.T(g["proj_used_poses"]).cuda()
    if cons:  # the reference's own draws of every iteration
        mp._consistency_draws = [(gio.T(g["cons_near_index"][it]), gio.T(g["cons_shift"][it])) for it in range(g["cons_shift"].shape[0])]
    if ba:
        mp.coord_pool = gio.T(g["ba_coord_pool"]).cuda()
        mp.used_poses = gio.T(g["ba_used_poses"]).cuda()
        mp.global_coord_pool = mp.global_coord_pool + 0.37  # stale, as after a bundle adjustment: only the batch ordering may look at it
        mp.ba_done_flag = True
    idx = gio.T(g["index_seq"]).to(torch.int64).cuda()
    iters = idx.shape[0]
    mp.mapping(iters, index_seq=idx)
    losses = mp.last_losses.cpu().numpy()
    for it in range(iters):
        assert abs(losses[it, 1] - g["loss_bce"][it]) <= 5e-6, (it, losses[it], g["loss_bce"][it])
        assert abs(losses[it, 0] - g["loss_total"][it]) <= 5e-6, (it, losses[it], g["loss_total"][it])
    last = iters - 1
    assert maxerr(nm.local_geo_features, g[f"it{last}_theta"]) <= 1e-4
    for n, t in zip(("W1", "b1", "W2", "b2"), dec.flat_params()):
        assert maxerr(t, g[f"it{last}_{n}"]) <= 1e-4, n
    assert cert_close(nm.local_point_certainties, g[f"it{last}_certainties"])
    assert np.array_equal(nm.local_point_ts_update.cpu().numpy(), g[f"it{last}_ts_update"])
    # rows the reference never touched keep their exact initial value
    init = gio.T(p["base_geo_features"])[gio.T(g["local_mask"])]
    touched = np.zeros(init.shape[0], bool)
    for it in range(iters):
        touched[g[f"it{it}_grad_theta_rows"].astype(np.int64)] = True
    assert torch.equal(nm.local_geo_features.detach().cpu()[~touched], init[~touched])
    # a10 write-back
    base = p["base_geo_features"].copy()
    base[g["final_geo_rows"].astype(np.int64)] = g["final_geo_vals"]
    assert maxerr(nm.geo_features, base) <= 1e-4
    assert cert_close(nm.point_certainties, g["final_point_certainties"])
    assert np.array_equal(nm.point_ts_update.cpu().numpy(), g["final_point_ts_update"])


def _fused_grads(env, cfg, p, g, index, batch_offset=0, n_main=None, n_eik=None, frozen=False, split=False,
                 variant=0, sdf_out=None):
    """Run clid_train_fwd_bwd (or, split=True, clid_train_search + clid_train_decode) once (no Adam) and
    return (grad buffer, loss[4], certainties, ts).  `variant` selects the decode kernel of the split form
    (include/clid_native.h clid_train_args.decode_variant): the tile kernels accumulate into 16-float rows whose column 8
    carries the certainty increments; the result is converted back to the compact layout here.  `sdf_out`
    (tile kernels): list that receives (records, sdf per record slot)."""
    import ctypes as C
    from clid_slam_amd import _lib

    lib = _lib.load()
    nm = env.neural_points(cfg, base=p)
    dec = env.decoder(cfg, g, "init_")
    mp, _ = env.mapper(cfg, nm, dec)
    bs = index.shape[0]
    decim = cfg.gradient_decimation
    view, keep = nm._map_view(True)
    tile = split and variant > 0
    n_rows = nm.local_geo_features.shape[0]
    if tile:
        grad = torch.zeros(_lib.GRAD_FEAT_OFFSET16 + n_rows * _lib.GRAD_ROW16, device="cuda")
    else:
        grad = torch.zeros(_lib.GRAD_FEAT_OFFSET + nm.local_geo_features.numel(), device="cuda")
    loss = torch.zeros(4, device="cuda")
    ws = torch.empty(int(lib.clid_train_workspace_floats(bs, decim, 1)), device="cuda")
    idx = index.to(torch.int64).cuda().contiguous()
    W1, b1, W2, b2 = dec.flat_params()
    ta = _lib.TrainArgs()
    ta.pool_coord, ta.pool_label = mp.global_coord_pool.data_ptr(), mp.sdf_label_pool.data_ptr()
    ta.pool_ts, ta.pool_weight = mp.time_pool.data_ptr(), mp.weight_pool.data_ptr()
    ta.index, ta.bs, ta.decimation, ta.batch_offset = idx.data_ptr(), bs, decim, batch_offset
    ta.fd_eps = float(cfg.voxel_size_m * cfg.num_grad_step_ratio)
    ta.inv_n_main = 1.0 / (n_main or bs)
    ta.inv_n_eik = 1.0 / (n_eik or ((bs + decim - 1) // decim))
    ta.sigma, ta.weight_e, ta.loss_weight_on, ta.eikonal_mode = float(mp.sdf_scale), 0.5, 1, 1
    ta.train_decoder = 0 if frozen else 1
    ta.W1, ta.b1, ta.W2, ta.b2 = W1.data_ptr(), b1.data_ptr(), W2.data_ptr(), b2.data_ptr()
    ta.sdf_scale, ta.defer_reduce = float(dec.sdf_scale), 0
    ta.grad, ta.ws, ta.loss_out = grad.data_ptr(), ws.data_ptr(), loss.data_ptr()
    ta.grad_stride = _lib.GRAD_ROW16 if tile else 0
    if split:
        rec = torch.empty(int(lib.clid_train_search_floats(bs, batch_offset, decim, 1, 1)), device="cuda")
        _lib.check(lib.clid_train_search(C.byref(view), C.byref(ta), 1, idx.data_ptr(), bs, rec.data_ptr(),
                                         _lib.stream()), "clid_train_search")
        ta.decode_variant, ta.pipeline = variant, 1
        sdf = None
        n_tasks = int(lib.clid_train_search_tasks(bs, batch_offset, decim, 1))
        if sdf_out is not None and tile:
            sdf = torch.zeros(n_tasks * 8, device="cuda")
            ta.sdf_dbg = sdf.data_ptr()
        assert lib.clid_train_decode_kernel(C.byref(view), C.byref(ta)) == variant
        _lib.check(lib.clid_train_decode(C.byref(view), C.byref(ta), rec.data_ptr(), _lib.stream()), "clid_train_decode")
        torch.cuda.synchronize()
        if sdf is not None:
            sdf_out.append((env.task_records(rec, 1, n_tasks)[0].cpu(), sdf.view(-1, 8).cpu()))
    else:
        _lib.check(lib.clid_train_fwd_bwd(C.byref(view), C.byref(ta), _lib.stream()), "clid_train_fwd_bwd")
    torch.cuda.synchronize()
    cert = nm.local_point_certainties.cpu()
    grad = grad.cpu()
    if tile:  # back to the compact layout: [836 | rows x 8], certainty increments out of column 8
        rows = grad[_lib.GRAD_FEAT_OFFSET16:].view(n_rows, _lib.GRAD_ROW16)
        cert = cert + rows[: cert.shape[0], 8]
        assert not rows[:, 9:].any()
        grad = torch.cat((grad[: _lib.GRAD_FEAT_OFFSET], rows[:, :8].reshape(-1)))
    return grad, loss.cpu(), cert, nm.local_point_ts_update.cpu()


@pytest.mark.parametrize("batch_offset", [0, 7])
def test_search_plus_decode_equals_the_fused_kernel(env, batch_offset):
    """clid_train_search + clid_train_decode (the hoisted-search form clid_mapping_run uses by default) and the
    single fused kernel of clid_train_fwd_bwd are the same computation: identical up to the order of the
    atomic accumulations."""
    g = gio.load("g6_loop_numerical_train_ln0.npz")
    p = gio.load("pool.npz")
    cfg = env.config(layer_norm_on=False)
    index = gio.T(g["index_seq"])[0]
    a = _fused_grads(env, cfg, p, g, index, batch_offset=batch_offset)
    b = _fused_grads(env, cfg, p, g, index, batch_offset=batch_offset, split=True)
    assert maxerr(a[0], b[0]) <= 2e-7 * max(1.0, float(a[0].abs().max()))
    assert maxerr(a[1], b[1]) <= 1e-6
    assert maxerr(a[2], b[2]) <= 1e-5
    assert torch.equal(a[3], b[3])


def test_mapping_loop_fused_schedule_matches_hoisted(env):
    """CLID_PIPELINE=0 schedule of clid_mapping_run (fused kernel per iteration) against the golden loop too."""
    from clid_slam_amd import _lib

    prev, _lib.PIPELINE = _lib.PIPELINE, 0  # the process default a Mapper without its own `pipeline` attribute uses
    try:
        test_mapping_loop_g6(env, "numerical", False, 0)
    finally:
        _lib.PIPELINE = prev


def test_fused_iteration_gradients_vs_reference(env):
    """The gradient buffer of ONE fused iteration against the reference's own gradients (G6, it0)."""
    from clid_slam_amd import _lib

    g = gio.load("g6_loop_numerical_train_ln0.npz")
    p = gio.load("pool.npz")
    cfg = env.config(bs=int(g["index_seq"].shape[1]))
    grad, loss, cert, ts = _fused_grads(env, cfg, p, g, gio.T(g["index_seq"])[0])
    assert abs(float(loss[1]) - float(g["loss_bce"][0])) <= 5e-6
    assert abs(float(loss[0]) - float(g["loss_total"][0])) <= 5e-6
    H, D = _lib.H, _lib.D
    parts = {"W1": grad[: H * D].view(H, D), "b1": grad[H * D : H * D + H],
             "W2": grad[H * D + H : H * D + 2 * H].view(1, H), "b2": grad[H * D + 2 * H : H * D + 2 * H + 1]}
    for n, t in parts.items():
        ref = g[f"it0_grad_{n}"]
        assert np.abs(t.numpy() - ref).max() <= 1e-4 * max(np.abs(ref).max(), 1e-12), n
    gt = grad[_lib.GRAD_FEAT_OFFSET :].view(-1, 8).numpy()
    rows = g["it0_grad_theta_rows"].astype(np.int64)
    dense = np.zeros_like(gt)
    dense[rows] = g["it0_grad_theta_vals"]
    assert np.abs(gt - dense).max() <= 1e-4 * np.abs(dense).max()
    untouched = np.ones(gt.shape[0], bool)
    untouched[rows] = False
    assert not gt[untouched].any()  # exact zeros stay exact (Adam eps = 1e-15 would amplify anything else)
    assert cert_close(cert, g["it0_certainties"])
    assert np.array_equal(ts.numpy(), g["it0_ts_update"])


@pytest.mark.parametrize("parts", [2, 4])
def test_sharded_gradients_sum_to_the_full_batch(env, parts):
    """Linearity at the full ncd128 batch size: shards of the global batch (batch_offset, global loss
    normalisers -- what each rank runs before the RCCL all-reduce) sum to the single-GPU result."""
    p = gio.load("pool.npz")
    g = gio.load("g6_loop_numerical_train_ln0.npz")
    bs = 16384
    cfg = env.config(bs=bs)
    gen = torch.Generator().manual_seed(3)
    index = torch.randint(0, p["coord"].shape[0], (bs,), generator=gen)
    full, loss_full, cert_full, ts_full = _fused_grads(env, cfg, p, g, index)
    z = gio.load("state.npz")
    cert0 = gio.T(p["base_point_certainties"])[gio.T(z["local_mask"])[:-1]]
    acc = torch.zeros_like(full)
    loss = torch.zeros(4)
    cert = torch.zeros_like(cert_full)
    ts = None
    n_eik = (bs + 9) // 10
    per = bs // parts
    for r in range(parts):
        gr, lo, ce, t = _fused_grads(env, cfg, p, g, index[r * per : (r + 1) * per], batch_offset=r * per,
                                     n_main=bs, n_eik=n_eik)
        acc += gr
        loss += lo
        cert += ce - cert0
        ts = t if ts is None else torch.maximum(ts, t)
    scale = float(full.abs().max())
    assert float((acc - full).abs().max()) <= 2e-5 * scale
    assert float((loss - loss_full).abs().max()) <= 2e-6
    assert cert_close(cert, cert_full - cert0)
    assert torch.equal(ts, ts_full)


def test_full_size_invariants(env):
    """Size-independent properties at bs = 16384 on the golden map: IDW weights sum to 1 wherever a
    neighbour exists, neighbours come out sorted by distance, the certainty mass added by one query pass
    equals the number of queries with a neighbour, and a zero learning rate leaves every parameter
    bit-identical."""
    p = gio.load("pool.npz")
    g = gio.load("g6_loop_numerical_train_ln0.npz")
    cfg = env.config(bs=16384)
    nm = env.neural_points(cfg, base=p)
    gen = torch.Generator().manual_seed(11)
    x = gio.T(p["coord"])[torch.randint(0, p["coord"].shape[0], (16384,), generator=gen)].cuda()
    c0 = nm.local_point_certainties.double().sum().item()
    f, _, w, nn, cert = nm.query_feature(x, None, training_mode=True)
    has = nn > 0
    ws = w.squeeze(-1).sum(1)
    assert float((ws[has] - 1).abs().max()) <= 1e-5 and float(ws[~has].abs().max()) == 0.0
    assert abs((nm.local_point_certainties.double().sum().item() - c0) - int(has.sum())) <= 0.05
    d2, idx = nm.radius_neighborhood_search(x[:2048])
    assert ((idx >= 0).sum(1) >= 0).all()
    wv = w.squeeze(-1)
    assert (wv[:, :-1] + 1e-12 >= wv[:, 1:]).all()  # ascending distance == descending weight
    dec = env.decoder(cfg, g, "init_")
    mp, _ = env.mapper(cfg, nm, dec)
    before = [t.detach().clone() for t in (nm.local_geo_features, *dec.flat_params())]
    cfg.lr = 0.0
    mp.mapping(2)
    for a, b in zip(before, (nm.local_geo_features, *dec.flat_params())):
        assert torch.equal(a, b.detach())


@pytest.mark.parametrize("mode,ln", [("analytic", 1), ("numerical", 1), ("analytic", 0)])
def test_mapping_loop_vs_oracle_free_batches(env, mode, ln):
    """Modes without a reference fixture (analytic + layer norm) and fresh random batches: the fused loop
    against the pinned CPU oracle, 3 iterations, bs 4096."""
    p = gio.load("pool.npz")
    g = gio.load("g6_loop_numerical_train_ln0.npz")
    bs, iters = 4096, 3
    cfg = env.config(layer_norm_on=bool(ln), bs=bs)
    if mode == "analytic":
        cfg.numerical_grad, cfg.gradient_decimation = False, 1
    gen = torch.Generator().manual_seed(5)
    idx = torch.randint(0, p["coord"].shape[0], (iters, bs), generator=gen)
    nm = env.neural_points(cfg, base=p)
    dec = env.decoder(cfg, g, "init_")
    mp, _ = env.mapper(cfg, nm, dec)
    mp.mapping(iters, index_seq=idx.cuda())
    st = gio.map_state(layer_norm_on=bool(ln))
    st.local_geo_features = gio.T(p["base_geo_features"])[gio.T(g["local_mask"])].clone()
    st.local_point_certainties = gio.T(p["base_point_certainties"])[gio.T(g["local_mask"])[:-1]].clone()
    st.local_point_ts_update = gio.T(p["base_point_ts_update"])[gio.T(g["local_mask"])[:-1]].clone()
    pool, _ = gio.sample_pool()
    od = gio.decoder(g, "init_")
    lc = O.LoopConfig(numerical_grad=(mode == "numerical"), gradient_decimation=10 if mode == "numerical" else 1)
    recs = O.mapping_iters(st, od, pool, idx, lc, record=True)
    got = mp.last_losses.cpu()
    for it, r in enumerate(recs):
        assert abs(float(got[it, 0]) - float(r["loss"])) <= 1e-5, (it, got[it], r["loss"])
        assert abs(float(got[it, 2]) - float(r["eikonal_loss"])) <= 1e-5
    assert maxerr(nm.local_geo_features, recs[-1]["theta"]) <= 1e-4
    for t, o in zip(dec.flat_params(), recs[-1]["dec"]):
        assert maxerr(t, o) <= 1e-4
    assert cert_close(nm.local_point_certainties, recs[-1]["certainties"])


@pytest.mark.parametrize("branch,ln", [("wf0_analytic", 0), ("wf0_analytic", 1), ("proj", 1), ("proj_no_eik", 0), ("cons", 1),
                                       ("sdf_l2", 1), ("zhong", 0), ("ba", 1)])
def test_loop_branches_vs_oracle_on_fresh_batches(env, branch, ln):
    """The reference's non-default loop branches (utils/mapper.py:646-658, 676-680, 695-696, 712-741, 751-776) away from their
    fixtures' batches: bs 8192, fresh random draws, 2 iterations, against the pinned CPU oracle -- losses, features, decoder,
    certainties."""
    from oracle import cpu_ref as OO

    p = gio.load("pool.npz")
    g = gio.load("g6_loop_numerical_train_ln0.npz")
    bs, iters = 8192, 2
    over = dict(layer_norm_on=bool(ln), bs=bs)
    lc_over = {}
    if branch == "wf0_analytic":
        over.update(weighted_first=False)
    elif branch in ("proj", "proj_no_eik"):
        over.update(proj_correction_on=True)
        lc_over.update(proj_correction_on=True)
        if branch == "proj_no_eik":  # the correction alone: g is still evaluated, the eikonal term is not
            over.update(ekional_loss_on=False)
            lc_over.update(ekional_loss_on=False)
    elif branch == "cons":
        over.update(consistency_loss_on=True)
        lc_over.update(consistency_loss_on=True, weight_c=0.5)
    elif branch in ("sdf_l2", "zhong"):
        over.update(main_loss_type=branch)
        lc_over.update(main_loss_type=branch)
    cfg = env.config(**over)
    if branch == "wf0_analytic":
        cfg.numerical_grad, cfg.gradient_decimation = False, 1
        lc_over.update(numerical_grad=False, gradient_decimation=1)
    if branch == "cons":
        cfg.consistency_count = 1024
    gen = torch.Generator().manual_seed(17)
    n_pool = p["coord"].shape[0]
    idx = torch.randint(0, n_pool, (iters, bs), generator=gen)
    nm = env.neural_points(cfg, base=p)
    dec = env.decoder(cfg, g, "init_")
    mp, _ = env.mapper(cfg, nm, dec)
    st = gio.map_state(layer_norm_on=bool(ln), weighted_first=branch != "wf0_analytic")
    st.local_geo_features = gio.T(p["base_geo_features"])[gio.T(g["local_mask"])].clone()
    st.local_point_certainties = gio.T(p["base_point_certainties"])[gio.T(g["local_mask"])[:-1]].clone()
    st.local_point_ts_update = gio.T(p["base_point_ts_update"])[gio.T(g["local_mask"])[:-1]].clone()
    pool, _ = gio.sample_pool()
    od = gio.decoder(g, "init_")
    if branch in ("sdf_l2", "zhong"):
        od.sdf_scale = 1.0  # model/decoder.py:51-53
    cons_seq = None
    poses = torch.eye(4, dtype=torch.float64)[None].repeat(3, 1, 1)
    for fi, spos in enumerate(((0.0, 0.0, 1.5), (6.0, 2.0, 1.5), (9.0, 3.0, 1.6))):
        poses[fi, :3, 3] = torch.tensor(spos, dtype=torch.float64)
    if branch in ("proj", "proj_no_eik"):
        mp.used_poses = poses.cuda()
        pool.frame_poses = poses
    if branch == "cons":
        cons_seq = [(torch.randint(0, bs, (1024,), generator=gen), torch.rand((bs, 3), generator=gen) * 2 * 0.05 - 0.05) for _ in range(iters)]
        mp._consistency_draws = cons_seq
    if branch == "ba":  # the pool in sensor frames under three rotated / shifted poses; the world-frame pool is stale
        gp = torch.Generator().manual_seed(29)
        for fi in range(3):
            a = float((torch.rand((), generator=gp, dtype=torch.float64) - 0.5) * 0.8)
            poses[fi, :3, :3] = torch.tensor([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]], dtype=torch.float64)
        P = poses[pool.time.long()]
        local = torch.bmm(P[:, :3, :3].transpose(1, 2), (pool.global_coord.double() - P[:, :3, 3]).unsqueeze(-1)).squeeze(-1).float()
        pool.local_coord, pool.used_poses = local, poses
        pool.global_coord = pool.global_coord + 0.4
        mp.coord_pool, mp.used_poses = local.cuda(), poses.cuda()
        mp.global_coord_pool = mp.global_coord_pool + 0.4
        mp.ba_done_flag = True
    mp.mapping(iters, index_seq=idx.cuda())
    lc = OO.LoopConfig(**lc_over)
    recs = OO.mapping_iters(st, od, pool, idx, lc, record=True, consistency_seq=cons_seq)
    got = mp.last_losses.cpu()
    for it, r in enumerate(recs):
        assert abs(float(got[it, 0]) - float(r["loss"])) <= 1e-5, (branch, it, got[it], r["loss"])
        assert abs(float(got[it, 1]) - float(r["sdf_loss"])) <= 1e-5
        assert abs(float(got[it, 2]) - float(r["eikonal_loss"])) <= 1e-5
        if branch == "cons":
            assert abs(float(got[it, 3]) - float(r["consistency_loss"])) <= 1e-5
    assert maxerr(nm.local_geo_features, recs[-1]["theta"]) <= 1e-4
    for t, o in zip(dec.flat_params(), recs[-1]["dec"]):
        assert maxerr(t, o) <= 1e-4
    assert cert_close(nm.local_point_certainties, recs[-1]["certainties"])
    assert np.array_equal(nm.local_point_ts_update.cpu().numpy(), recs[-1]["ts_update"].numpy())


@pytest.mark.parametrize("pipeline,ln,eik,train", [(1, 0, True, True), (1, 1, True, True), (1, 0, False, True), (1, 1, True, False),
                                                   (0, 0, True, True)])
def test_mapping_weighted_first_false_vs_oracle(env, pipeline, ln, eik, train):
    """`weighted_first: False` (every neighbour decoded, SDFs blended; utils/mapper.py:679-680) against the oracle: the
    fused iteration of csrc/train_wf0.hip on the hoisted schedule (pipeline 1: numerical / no eikonal term, layer norm,
    frozen decoder) and the un-fused loop over the HIP autograd ops it replaced (pipeline 0)."""
    p = gio.load("pool.npz")
    g = gio.load(f"g6_loop_numerical_train_ln{ln}.npz")
    bs, iters = 2048, 3
    cfg = env.config(weighted_first=False, layer_norm_on=bool(ln), bs=bs, ekional_loss_on=eik)
    gen = torch.Generator().manual_seed(9)
    idx = torch.randint(0, p["coord"].shape[0], (iters, bs), generator=gen)
    nm = env.neural_points(cfg, base=p)
    dec = env.decoder(cfg, g, "init_")
    if not train:
        for q in dec.parameters():
            q.requires_grad_(False)
    mp, _ = env.mapper(cfg, nm, dec)
    mp.pipeline = pipeline
    mp.mapping(iters, index_seq=idx.cuda())
    st = gio.map_state(weighted_first=False, layer_norm_on=bool(ln))
    m = gio.T(g["local_mask"])
    st.local_geo_features = gio.T(p["base_geo_features"])[m].clone()
    st.local_point_certainties = gio.T(p["base_point_certainties"])[m[:-1]].clone()
    st.local_point_ts_update = gio.T(p["base_point_ts_update"])[m[:-1]].clone()
    pool, _ = gio.sample_pool()
    od = gio.decoder(g, "init_")
    recs = O.mapping_iters(st, od, pool, idx, O.LoopConfig(ekional_loss_on=eik, train_decoder=train), record=True)
    got = mp.last_losses.cpu()
    for it, r in enumerate(recs):
        assert abs(float(got[it, 0]) - float(r["loss"])) <= 1e-5, (it, got[it], r["loss"])
        if eik:
            assert abs(float(got[it, 2]) - float(r["eikonal_loss"])) <= 1e-5
    assert maxerr(nm.local_geo_features, recs[-1]["theta"]) <= 1e-4
    for t, o in zip(dec.flat_params(), recs[-1]["dec"]):
        assert maxerr(t, o) <= (1e-4 if train else 0.0)
    assert cert_close(nm.local_point_certainties, recs[-1]["certainties"])
    assert torch.equal(nm.local_point_ts_update.cpu(), st.local_point_ts_update)


def test_default_buffer_size_scene_vs_oracle(env):
    """The shipped configuration end to end (buffer_size = 5e7, 128x1024-ray synthetic scan, map built by
    the product's own NeuralPoints.update on the GPU): 5 free-running fused iterations at bs = 16384 against
    the CPU oracle on the same state.  Exercises the exact fp64 hash modulo and int32 slot arithmetic at the
    real table size."""
    import bench
    from clid_slam_amd import HotPathConfig

    cfg = HotPathConfig()
    cfg.device = "cuda:0"
    nm, dec, mp, scene = bench.build_scene(cfg, "cuda:0")
    gen = torch.Generator().manual_seed(77)
    iters, bs = 5, cfg.bs
    idx = torch.randint(0, mp.pool_sample_count, (iters, bs), generator=gen)
    cpu = lambda t: t.detach().cpu().clone()
    dx, mvd = O.search_neighborhood(cfg.num_nei_cells, cfg.search_alpha, cfg.voxel_size_m)
    st = O.MapState(
        buffer_pt_index=cpu(nm.buffer_pt_index), neural_points=cpu(nm.neural_points),
        point_ts_create=cpu(nm.point_ts_create), travel_dist=cpu(nm.travel_dist), cur_ts=int(nm.cur_ts),
        global2local=cpu(nm.global2local), local_neural_points=cpu(nm.local_neural_points),
        local_geo_features=cpu(nm.local_geo_features.data), local_point_certainties=cpu(nm.local_point_certainties),
        local_point_ts_update=cpu(nm.local_point_ts_update), resolution=cfg.voxel_size_m, buffer_size=cfg.buffer_size,
        diff_travel_dist_local=nm.diff_travel_dist_local, neighbor_dx=dx, max_valid_dist2=mvd)
    od = O.DecoderParams(*[cpu(p) for p in dec.flat_params()], sdf_scale=dec.sdf_scale)
    pool = O.SamplePool(cpu(mp.global_coord_pool), cpu(mp.sdf_label_pool), cpu(mp.time_pool), cpu(mp.weight_pool))
    # the search itself, bit-exact, on a slice of the first batch
    x = pool.global_coord[idx[0, :4096]]
    d2o, io = O.radius_neighborhood_search(st, x)
    d2h, ih = nm.radius_neighborhood_search(x.cuda())
    assert torch.equal(ih.cpu(), io) and torch.equal(d2h.cpu(), d2o)
    mp.mapping(iters, index_seq=idx.cuda())
    recs = O.mapping_iters(st, od, pool, idx, O.LoopConfig(sigma=mp.sdf_scale), record=True)
    got = mp.last_losses.cpu()
    for it, r in enumerate(recs):
        assert abs(float(got[it, 0]) - float(r["loss"])) <= 2e-5, (it, got[it], r["loss"])
    assert maxerr(nm.local_geo_features, recs[-1]["theta"]) <= 1e-4
    for t, o in zip(dec.flat_params(), recs[-1]["dec"]):
        assert maxerr(t, o) <= 1e-4
    assert cert_close(nm.local_point_certainties, recs[-1]["certainties"])


@pytest.mark.parametrize("bs", [65536, 262144])
def test_sharded_gradients_at_large_batches(env, bs):
    """BASELINE configs 2 and 3 batch sizes (65 536 and 262 144 samples per iteration, the latter "sharded 4x"):
    4 shards through the C ABI -- fused kernel and the search + decode pair -- sum to the full batch."""
    p = gio.load("pool.npz")
    g = gio.load("g6_loop_numerical_train_ln0.npz")
    cfg = env.config(bs=bs)
    gen = torch.Generator().manual_seed(4)
    index = torch.randint(0, p["coord"].shape[0], (bs,), generator=gen)
    full, loss_full, _, _ = _fused_grads(env, cfg, p, g, index)
    acc = torch.zeros_like(full)
    loss = torch.zeros(4)
    per = bs // 4
    for r in range(4):
        gr, lo, _, _ = _fused_grads(env, cfg, p, g, index[r * per : (r + 1) * per], batch_offset=r * per, n_main=bs,
                                    n_eik=(bs + 9) // 10, split=(r % 2 == 1))
        acc += gr
        loss += lo
    assert float((acc - full).abs().max()) <= 5e-5 * float(full.abs().max())
    assert float((loss - loss_full).abs().max()) <= 2e-6


@pytest.mark.parametrize("ln,wf,tile", [(0, 1, 0), (1, 1, 0), (0, 0, 0), (1, 0, 0), (0, 1, 1), (1, 1, 1)])
def test_tracking_measurement_model_g8(env, ln, wf, tile, monkeypatch):
    """Row N1: fused h_model kernel (per-point outputs and the float64 normal equations) against the
    reference's IEKFOM.h_model fixture, for both `weighted_first` settings (wf = 0: every neighbour decoded, SDFs
    blended, the std mask of utils/error_state_iekf.py:217-241 with a threshold inside the std's range)."""
    from clid_slam_amd import tracking

    # tile = 1: the matrix-core tile kernel (csrc/track_tile.hip; the default for scans beyond 12 288 points, forced here);
    # 0: the 16-lane kernel
    monkeypatch.setenv("CLID_TRACK_TILE", str(tile))
    g = gio.load("g8_tracking.npz")
    cfg = env.config(layer_norm_on=bool(ln), weighted_first=bool(wf))
    cfg.reg_min_grad_norm, cfg.reg_max_grad_norm = float(g["grad_window"][0]), float(g["grad_window"][1])
    if not wf:
        cfg.max_sdf_std_ratio = float(g["max_sdf_std_ratio_wf0"][ln])
    tag = f"ln{ln}" + ("" if wf else "_wf0")
    nm = env.neural_points(cfg)
    dec = env.decoder(cfg)
    rot, pos, pc = gio.T(g["rot"]), gio.T(g["pos"]), gio.T(g["pc_imu"]).cuda()
    z, H, vp, r_inv = tracking.h_model(nm, dec, cfg, rot, pos, pc)
    Hr, zr, rr = g[f"H6_{tag}"], g[f"z_{tag}"], g[f"R_inv_{tag}"]
    assert H.dtype == torch.float64 and H.shape == (Hr.shape[0], 18)
    assert maxerr(z, zr) <= 2e-6
    assert maxerr(H[:, :6], Hr) <= 5e-5 and float(H[:, 6:].abs().max()) == 0.0
    assert maxerr(vp, g[f"valid_points_{tag}"]) <= 1e-5
    assert maxerr(r_inv, rr) <= 2e-2
    S, b, n = tracking.normal_equations(nm, dec, cfg, rot, pos, pc)
    assert n == Hr.shape[0]
    S_ref = (Hr.T * rr) @ Hr
    b_ref = (Hr.T * rr) @ zr
    assert np.abs(S[:6, :6].cpu().numpy() - S_ref).max() <= 2e-4 * np.abs(S_ref).max()
    assert np.abs(b[:6].cpu().numpy() - b_ref).max() <= 2e-4 * max(np.abs(b_ref).max(), 1e-9)
    assert float(S[6:, :].abs().max()) == 0.0 and float(S[:, 6:].abs().max()) == 0.0


@pytest.mark.parametrize("n,ln", [(13000, False), (30000, False), (30000, True), (777, True)])
def test_tracking_tile_kernel_equals_the_16_lane_kernel(n, ln, monkeypatch):
    """The tile form of the measurement model (csrc/track_tile.hip: 8 points per wave up to 24 576 points, 16 beyond; ragged last
    tile) against the 16-lane kernel the G8 fixtures pin, on scans of the sizes that select it: SDF, gradient, mapped points and
    validity of every point, and the float64 normal equations."""
    import bench
    from clid_slam_amd import HotPathConfig, tracking

    cfg = HotPathConfig()
    cfg.device, cfg.layer_norm_on = "cuda:0", ln
    cfg.reg_min_grad_norm, cfg.reg_max_grad_norm = 1e-3, 1e3  # (random features: the mask is decided by the neighbour count)
    nm, dec, mp, scene = bench.build_scene(cfg, "cuda:0")
    g = torch.Generator().manual_seed(n)
    pick = torch.randint(0, scene["coord"].shape[0], (n,), generator=g)
    sensor = scene["sensor"].to(torch.float32)
    pc = (scene["coord"][pick] + 0.05 * torch.randn((n, 3), generator=g) - sensor).cuda().contiguous()
    pc[:5] += 40.0  # a few points far from the map (no neighbours)
    rot = torch.tensor([[0.9998, -0.02, 0.0], [0.02, 0.9998, 0.0], [0.0, 0.0, 1.0]])
    pos = sensor + torch.tensor([0.01, -0.02, 0.005])
    res = {}
    for tile in ("0", "1"):
        monkeypatch.setenv("CLID_TRACK_TILE", tile)
        x, out, ne = tracking._launch(nm, dec, cfg, rot, pos, pc, True, True)
        torch.cuda.synchronize()
        res[tile] = ({k: v.clone() for k, v in out.items()}, ne.clone().sum(0)[:28])
    a, b = res["0"][0], res["1"][0]
    assert maxerr(a["sdf"], b["sdf"]) <= 2e-6 and maxerr(a["pmap"], b["pmap"]) == 0.0
    assert maxerr(a["grad"], b["grad"]) <= 5e-5 * max(1.0, float(a["grad"].abs().max()))
    differ = a["valid"] != b["valid"]  # only where the gradient norm sits on a threshold of the mask
    gn = a["grad"].norm(dim=1)
    edge = ((gn - cfg.reg_min_grad_norm).abs() < 1e-4) | ((gn - cfg.reg_max_grad_norm).abs() < 1e-4)
    assert not bool((differ & ~edge).any()) and int(a["valid"].sum()) > n // 4
    if not bool(differ.any()):
        sa, sb = res["0"][1], res["1"][1]
        assert float((sa - sb).abs().max()) <= 1e-6 * float(sa.abs().max()) and sa[27] == sb[27] == float(a["valid"].sum())


def test_tracking_model_on_an_empty_scan_keeps_the_reduction_ring_consistent(env):
    """A scan without points: `normal_equations` returns zeros and n_valid 0 instead of waiting for an epoch nobody
    publishes, and the evaluations around it are unaffected -- the ring buffer the NEXT evaluation accumulates into is still
    cleared by the empty one (the reference's h_model on an empty tensor returns empty outputs, utils/error_state_iekf.py:176)."""
    from clid_slam_amd import tracking

    g = gio.load("g8_tracking.npz")
    cfg = env.config(layer_norm_on=False)
    cfg.reg_min_grad_norm, cfg.reg_max_grad_norm = float(g["grad_window"][0]), float(g["grad_window"][1])
    nm, dec = env.neural_points(cfg), env.decoder(cfg)
    rot, pos, pc = gio.T(g["rot"]), gio.T(g["pos"]), gio.T(g["pc_imu"]).cuda()
    empty = torch.empty((0, 3), device="cuda:0")
    S0, b0, n0 = tracking.normal_equations(nm, dec, cfg, rot, pos, pc)
    for _ in range(4):  # more evaluations than the ring has buffers, empty and full interleaved
        Se, be, ne = tracking.normal_equations(nm, dec, cfg, rot, pos, empty)
        assert ne == 0 and float(Se.abs().max()) == 0.0 and float(be.abs().max()) == 0.0
        S1, b1, n1 = tracking.normal_equations(nm, dec, cfg, rot, pos, pc)
        # (float64 atomics: two evaluations differ in the last bits; a buffer that had not been cleared would double the sums)
        assert n1 == n0 and torch.allclose(S1, S0, rtol=1e-10, atol=0.0) and torch.allclose(b1, b0, rtol=1e-10, atol=1e-14)
    z, H, vp, r_inv = tracking.h_model(nm, dec, cfg, rot, pos, empty)
    assert z.shape == (0,) and H.shape == (0, 18) and vp.shape == (0, 3) and r_inv.shape == (0,)


def test_iekfom_h_model_method_form_g8(env):
    """Row N1 through the METHOD form a maintainer patches onto the reference's IEKFOM (same attributes read of `self`,
    `self.R_inv` stored, 3-tuple returned) against the reference's own h_model output (G8)."""
    from clid_slam_amd.tracking import IEKFOMMeasurement

    g = gio.load("g8_tracking.npz")
    cfg = env.config(layer_norm_on=False)
    cfg.reg_min_grad_norm, cfg.reg_max_grad_norm = float(g["grad_window"][0]), float(g["grad_window"][1])

    class State:
        rot, pos = gio.T(g["rot"]), gio.T(g["pos"])

    class Filter(IEKFOMMeasurement):
        pass

    f = Filter()
    f.config, f.neural_points, f.geo_decoder, f.x, f.R_inv = cfg, env.neural_points(cfg), env.decoder(cfg), State(), None
    z, H, pts = f.h_model(gio.T(g["pc_imu"]).cuda())
    assert H.shape[1] == 18 and pts.shape[1] == 3 and f.R_inv is not None and f.R_inv.shape == z.shape
    assert z.shape[0] == g["z_ln0"].shape[0]
    assert maxerr(z, g["z_ln0"]) <= 2e-6 and maxerr(f.R_inv, g["R_inv_ln0"]) <= 2e-2 and maxerr(H[:, :6], g["H6_ln0"]) <= 5e-5


@pytest.mark.parametrize("ln,loc", [(0, 0), (1, 0), (0, 1)])
def test_dense_sdf_query_vs_oracle(env, ln, loc):
    """Row N3: the fused dense query (global map, no time filter) against the oracle's
    query_feature -> decoder composition (both pinned on G2/G3), incl. the nn >= 1 / nn >= 4 masks."""
    from clid_slam_amd import mesher

    g = gio.load("g2_query.npz")
    cfg = env.config(layer_norm_on=bool(ln))
    nm = env.neural_points(cfg)
    dec = env.decoder(cfg)
    gen = torch.Generator().manual_seed(1)
    x = torch.cat([gio.T(g["x"]), gio.T(g["x"]) + 0.3 * torch.randn(g["x"].shape, generator=gen)])
    sdf, _, _, mask = mesher.query_points(nm, dec, cfg, x.cuda(), bs=700, query_locally=bool(loc), mask_min_nn_count=4)
    st = gio.map_state(layer_norm_on=bool(ln))
    f, _, nn, _, _ = O.query_feature(st, x, training_mode=False, query_locally=bool(loc))
    ref = torch.where(nn >= 1, O.mlp_sdf(gio.decoder(), f), torch.zeros(()))
    assert maxerr(sdf, ref) <= 2e-6
    assert torch.equal(mask.cpu().bool(), nn >= 4)
    assert (nn == 0).any() and (nn >= 4).any()


@pytest.mark.parametrize("ln,loc", [(0, 0), (0, 1), (1, 0), (1, 1)])
def test_sdf_query_tile_kernel_equals_the_16_lane_kernel(env, ln, loc, monkeypatch):
    """Row N3: the matrix-core tile kernel (csrc/query_tile.hip: 8-lane search of the training launches + MFMA decoder) against
    the 16-lane VALU kernel it replaces, on points inside the map, around it (outside the cell directory's box: probing
    search) and far away; ragged sizes.  Neighbour counts bit-exact, SDF within fp32 summation-order noise."""
    from clid_slam_amd import mesher

    g = gio.load("g2_query.npz")
    cfg = env.config(layer_norm_on=bool(ln))
    nm = env.neural_points(cfg)
    dec = env.decoder(cfg)
    gen = torch.Generator().manual_seed(3)
    base = gio.T(g["x"])
    x = torch.cat([base, base + 0.3 * torch.randn(base.shape, generator=gen), base + 6.0 * torch.randn(base.shape, generator=gen),
                   base[:37] * 50.0]).cuda()
    for n in (x.shape[0], 1, 15, 17, 1000 + 13):
        out = {}
        for mode in ("0", "1"):
            monkeypatch.setenv("CLID_SDF_TILE", mode)
            sdf, _, _, mask = mesher.query_points(nm, dec, cfg, x[:n], bs=1 << 20, query_locally=bool(loc), mask_min_nn_count=4)
            out[mode] = (sdf.clone(), mask.clone())
        assert torch.equal(out["0"][1], out["1"][1]), n
        assert maxerr(out["0"][0], out["1"][0]) <= 1e-6, n
    assert out["1"][1].any() and not out["1"][1].all()


@pytest.mark.parametrize("ln,loc,wf", [(0, 0, 1), (0, 1, 1), (1, 0, 1), (1, 1, 1), (0, 0, 0), (0, 1, 0), (1, 0, 0), (1, 1, 0)])
def test_mesher_query_points_vs_reference_g12(env, ln, loc, wf):
    """Row N3 against the reference's own `Mesher.query_points` output (fixture G12), through the `Mesher` drop-in, for both
    `weighted_first` settings (utils/mesher.py:130-138)."""
    from clid_slam_amd.mesher import Mesher

    g = gio.load("g12_mesher.npz")
    cfg = env.config(layer_norm_on=bool(ln), weighted_first=bool(wf))
    tag = f"ln{ln}_loc{loc}" + ("" if wf else "_wf0")
    nm = env.neural_points(cfg)
    dec = env.decoder(cfg)
    mesher = Mesher(cfg, nm, {"sdf": dec, "semantic": None, "color": None})
    sdf, sem, col, mask = mesher.query_points(gio.T(g["x"]).cuda(), 700, query_locally=bool(loc), mask_min_nn_count=4, out_torch=True)
    assert sem is None and col is None
    assert maxerr(sdf, g[f"sdf_{tag}"]) <= 2e-6
    assert np.array_equal(mask.cpu().numpy().astype(np.uint8), g[f"mask_{tag}"])
    sdf_np, _, _, mask_np = mesher.query_points(gio.T(g["x"]).cuda(), 10_000, query_locally=bool(loc), out_torch=False)
    assert isinstance(sdf_np, np.ndarray) and maxerr(sdf_np, g[f"sdf_{tag}"]) <= 2e-6


@pytest.mark.parametrize("ln", [0, 1])
def test_query_sdf_and_gradient_neighbour_first_vs_oracle(env, ln):
    """`weighted_first: False` through NeuralPoints.query_sdf_and_gradient (the fused inference kernel): SDF and analytic
    d SDF / d x against the oracle's autograd through query_feature -> per-neighbour decode -> blend."""
    g = gio.load("g2_query.npz")
    cfg = env.config(layer_norm_on=bool(ln), weighted_first=False)
    nm = env.neural_points(cfg)
    dec = env.decoder(cfg)
    x = gio.T(g["x"])
    sdf, grad, nn, cert = nm.query_sdf_and_gradient(dec, x.cuda())
    st = gio.map_state(layer_norm_on=bool(ln), weighted_first=False)
    xr = x.clone().requires_grad_(True)
    f, w, nn_ref, cert_ref, _ = O.query_feature(st, xr, training_mode=False)
    s = (O.mlp_sdf(gio.decoder(), f) * w).sum(dim=1).squeeze(1)
    gr = torch.autograd.grad(s.sum(), xr)[0]
    assert maxerr(sdf, s.detach()) <= 2e-6 and maxerr(grad, gr) <= 5e-5
    assert torch.equal(nn.cpu(), nn_ref) and maxerr(cert, cert_ref) <= 1e-5
    assert float(gr.abs().max()) > 1e-3


@pytest.mark.parametrize("cells,alpha", [(1, 0.0), (2, 0.5)])
def test_query_certainty_direct_probe_vs_oracle(env, cells, alpha):
    """NeuralPoints.query_certainty (global map, model/neural_points.py:1032-1051): the direct probe of buffer_pt_index
    against the oracle, for the 1-cell neighbourhood process_frame uses (utils/mapper.py:409-423) and the default one."""
    g = gio.load("g2_query.npz")
    cfg = env.config()
    nm = env.neural_points(cfg)
    gen = torch.Generator().manual_seed(4)
    x = torch.cat([gio.T(g["x"]), gio.T(g["x"]) + 0.5 * torch.randn(g["x"].shape, generator=gen)])
    nm.set_search_neighborhood(num_nei_cells=cells, search_alpha=alpha)
    got = nm.query_certainty(x.cuda())
    st = gio.map_state()
    st.neighbor_dx, st.max_valid_dist2 = O.search_neighborhood(cells, alpha, st.resolution)
    ref = O.query_certainty(st, x)
    assert maxerr(got, ref) == 0.0
    assert (ref > 0).any() and (ref == 0).any()


def test_search_records_short_lists_equal_full_depth():
    """The chunked search keeps 3 candidates per lane and repeats a wave at full depth when one it pushed out
    could have been a winner; forcing the full-depth path for every wave (debug bit 2) must give bit-identical
    records, at the full batch size and over several iterations, with and without the probe prefilter."""
    import ctypes as C
    import bench
    from clid_slam_amd import HotPathConfig, _lib

    lib = _lib.load()
    cfg = HotPathConfig()
    cfg.device = "cuda:0"
    nm, dec, mp, scene = bench.build_scene(cfg, "cuda:0")
    bs, iters, decim = 16384, 3, cfg.gradient_decimation
    idx = torch.randint(0, mp.pool_sample_count, (iters, bs), device="cuda", generator=torch.Generator("cuda").manual_seed(3))
    view, keep = nm._map_view(True)
    ta = _lib.TrainArgs()
    ta.pool_coord, ta.pool_label = mp.global_coord_pool.data_ptr(), mp.sdf_label_pool.data_ptr()
    ta.pool_ts, ta.pool_weight = mp.time_pool.data_ptr(), mp.weight_pool.data_ptr()
    ta.bs, ta.decimation, ta.batch_offset, ta.eikonal_mode, ta.loss_weight_on = bs, decim, 0, 1, 1
    ta.fd_eps = float(cfg.voxel_size_m * cfg.num_grad_step_ratio)
    n = int(lib.clid_train_search_floats(bs, 0, decim, 1, iters))
    recs = []
    for flags in (0, 4):
        ta.debug_flags = flags
        rec = torch.zeros(n, device="cuda")
        _lib.check(lib.clid_train_search(C.byref(view), C.byref(ta), iters, idx.data_ptr(), bs, rec.data_ptr(),
                                         _lib.stream()), "clid_train_search")
        torch.cuda.synchronize()
        import shim_io
        recs.append(shim_io.task_records(rec, iters, int(lib.clid_train_search_tasks(bs, 0, decim, 1))).reshape(-1, 48, 4).cpu())
    a, b = recs
    assert torch.equal(a[:, :16].contiguous().view(torch.int32), b[:, :16].contiguous().view(torch.int32))  # positions, descriptors, labels (bit patterns)
    wa, wb = a[:, 16:].reshape(-1, 8, 8, 2)[:, :, :6], b[:, 16:].reshape(-1, 8, 8, 2)[:, :, :6]
    assert torch.equal(wa.contiguous().view(torch.int32), wb.contiguous().view(torch.int32))
    found = (wa[..., 1].contiguous().view(torch.int32) >= 0).float().mean()
    assert found > 0.5  # the comparison is not vacuous


SWEEP = [
    # (bs, decimation, num_nei_cells, search_alpha, layer_norm, loss_weight_on, eikonal)
    (37, 10, 2, 0.5, 0, True, True),      # tiny ragged batch: one partial lattice block + tail task
    (1000, 3, 2, 0.5, 0, True, True),     # sparsest task packing (2 tasks per 3 samples)
    (1001, 2, 2, 0.5, 1, True, True),     # every second sample on the lattice, odd batch, layer norm
    (513, 7, 1, 1.0, 0, False, True),     # 27-cell neighbourhood (P = 27), unweighted loss
    (640, 10, 3, 0.2, 0, True, True),     # 147-cell neighbourhood (P > 96: two probe chunks of the 16-lane search too)
    (777, 10, 2, 0.5, 0, True, False),    # eikonal term off: plain tasks only
    (256, 1, 2, 0.5, 0, True, True),      # numerical eikonal on EVERY sample (bundles only)
]


@pytest.mark.parametrize("bs,decim,nnc,alpha,ln,lw,eik", SWEEP)
def test_mapping_loop_configuration_sweep_vs_oracle(env, bs, decim, nnc, alpha, ln, lw, eik):
    """The fused loop against the pinned CPU oracle across batch shapes, decimations and search neighbourhoods
    the golden fixtures do not cover (2 iterations, fresh random batches)."""
    p = gio.load("pool.npz")
    g = gio.load("g6_loop_numerical_train_ln0.npz")
    iters = 2
    cfg = env.config(layer_norm_on=bool(ln), bs=bs, gradient_decimation=decim, num_nei_cells=nnc, search_alpha=alpha,
                     loss_weight_on=lw, ekional_loss_on=eik)
    gen = torch.Generator().manual_seed(1000 + bs)
    idx = torch.randint(0, p["coord"].shape[0], (iters, bs), generator=gen)
    nm = env.neural_points(cfg, base=p)
    nm.set_search_neighborhood(num_nei_cells=nnc, search_alpha=alpha)
    dec = env.decoder(cfg, g, "init_")
    mp, _ = env.mapper(cfg, nm, dec)
    mp.mapping(iters, index_seq=idx.cuda())
    st = gio.map_state(layer_norm_on=bool(ln))
    st.neighbor_dx, st.max_valid_dist2 = O.search_neighborhood(nnc, alpha, cfg.voxel_size_m)
    st.local_geo_features = gio.T(p["base_geo_features"])[gio.T(g["local_mask"])].clone()
    st.local_point_certainties = gio.T(p["base_point_certainties"])[gio.T(g["local_mask"])[:-1]].clone()
    st.local_point_ts_update = gio.T(p["base_point_ts_update"])[gio.T(g["local_mask"])[:-1]].clone()
    pool, _ = gio.sample_pool()
    od = gio.decoder(g, "init_")
    lc = O.LoopConfig(numerical_grad=True, gradient_decimation=decim, loss_weight_on=lw, ekional_loss_on=eik)
    recs = O.mapping_iters(st, od, pool, idx, lc, record=True)
    got = mp.last_losses.cpu()
    for it, r in enumerate(recs):
        assert abs(float(got[it, 0]) - float(r["loss"])) <= 2e-5, (it, got[it], r["loss"])
    assert maxerr(nm.local_geo_features, recs[-1]["theta"]) <= 1e-4
    for t, o in zip(dec.flat_params(), recs[-1]["dec"]):
        assert maxerr(t, o) <= 1e-4
    assert cert_close(nm.local_point_certainties, recs[-1]["certainties"])
    assert torch.equal(nm.local_point_ts_update.cpu(), recs[-1]["ts_update"])


def test_tiny_map_and_empty_inputs(env):
    """Edge cases: a local map with fewer points than K, queries with no neighbour at all, zero-length inputs."""
    from clid_slam_amd import Decoder, HotPathConfig, Mapper, NeuralPoints

    cfg = HotPathConfig()
    cfg.device, cfg.bs, cfg.bs_new_sample, cfg.buffer_size = "cuda", 16, 0, 100003
    torch.manual_seed(1)
    nm = NeuralPoints(cfg)
    nm.travel_dist = torch.zeros(2, device="cuda")
    pts = torch.tensor([[0.1, 0.1, 0.1], [0.5, 0.1, 0.1], [0.1, 0.9, 0.1]], device="cuda")
    nm.update(pts, torch.zeros(3, device="cuda"), torch.eye(3, device="cuda"), 0)
    assert nm.count() == 3 and nm.local_count() == 3
    nm.geo_features = torch.cat((0.1 * torch.randn(3, 8), torch.zeros(1, 8))).cuda()
    nm.reset_local_map(torch.zeros(3, device="cuda"), torch.eye(3, device="cuda"), 0, reboot_map=True)
    # queries: near the points, and far away (no neighbour in the 81 cells)
    x = torch.tensor([[0.2, 0.2, 0.2], [30.0, 30.0, 30.0], [0.4, 0.5, 0.0]], device="cuda")
    feat, _, w, nn_cnt, cert = nm.query_feature(x, training_mode=False)
    assert nn_cnt.tolist()[0] == 3 and nn_cnt.tolist()[1] == 0
    assert torch.isfinite(feat).all() and float(w[1].abs().sum()) == 0.0
    assert abs(float(w[0].sum()) - 1.0) <= 1e-5
    # zero-length query
    f0, _, w0, n0, _ = nm.query_feature(torch.empty((0, 3), device="cuda"), training_mode=False)
    assert f0.shape[0] == 0 and w0.shape[0] == 0 and n0.shape[0] == 0
    # a short training run on a 40-sample pool (every batch index repeats samples)
    dec = Decoder(cfg, cfg.geo_mlp_hidden_dim, cfg.geo_mlp_level, 1)

    class DS:
        lose_track = False
        stop_status = False
        processed_frame = 0
        gt_pose_provided = False

    mp = Mapper(cfg, DS(), nm, None, dec)
    g = torch.Generator().manual_seed(2)
    coord = torch.rand((40, 3), generator=g) * torch.tensor([1.0, 1.2, 0.4])
    coord[::7] += 25.0  # some samples far from the map
    mp.set_pool(coord, 0.1 * torch.randn(40, generator=g), torch.ones(40), torch.zeros(40, dtype=torch.int32))
    before = nm.local_geo_features.detach().clone()
    mp.mapping(3)
    assert torch.isfinite(mp.last_losses).all() and torch.isfinite(nm.local_geo_features).all()
    assert not torch.equal(before, nm.local_geo_features.detach())


@pytest.mark.parametrize("wide", [False, True])
def test_mapping_prep_resets_and_draws_like_the_host_restatement(wide, monkeypatch):
    """clid_mapping_prep: one launch zeroes the loop's workspace and draws [iters, bs] batch indices composed as
    utils/mapper.py:473-500 (history part uniform over the pool, the last bs_new columns picked from new_idx); every
    entry equals the host restatement of the generator; Mapper.mapping uses it and advances the call counter."""
    import ctypes as C

    from clid_slam_amd import _lib

    # wide: the ordering blocks' 64-bit composites everywhere (the path of key ranges whose kept keys lie more than 2^21 apart;
    # the default picks per block: 32-bit (key - smallest kept key, slot) composites when they fit)
    monkeypatch.setenv("CLID_SORT_WIDE", "1" if wide else "0")
    lib = _lib.load()
    dev = "cuda:0"
    iters, bs, bs_new, pool = 7, 1000, 300, 897_123
    new_idx = torch.arange(800_000, 800_000 + 5000, device=dev, dtype=torch.int64) * 1 + 17
    flat = torch.full((4096 + 8,), 3.0, device=dev)
    idx = torch.full((iters, bs), -5, device=dev, dtype=torch.int64)
    seed, counter = 42, 9
    _lib.check(lib.clid_mapping_prep(flat.data_ptr(), 4096, idx.data_ptr(), iters, bs, bs_new, pool, new_idx.data_ptr(),
                                     new_idx.shape[0], seed, counter, None, 0.4, None, 0, 0, 1, _lib.stream()), "clid_mapping_prep")
    torch.cuda.synchronize()
    assert float(flat[:4096].abs().max()) == 0.0 and float(flat[4096:].min()) == 3.0
    got = idx.cpu().numpy()
    nid = new_idx.cpu().numpy()
    for it, col in ((0, 0), (0, 699), (3, 17), (6, 699)):
        assert got[it, col] == lib.clid_debug_prep_draw(seed, counter, it * bs + col, pool)
    for it, col in ((0, 700), (2, 999), (6, 850)):
        assert got[it, col] == nid[lib.clid_debug_prep_draw(seed, counter, it * bs + col, nid.shape[0])]
    assert got[:, :700].min() >= 0 and got[:, :700].max() < pool and np.isin(got[:, 700:], nid).all()
    assert abs(got[:, :700].mean() / pool - 0.5) < 0.02 and len(np.unique(got[:, :700])) > 0.99 * 4900
    idx2 = torch.empty_like(idx)
    _lib.check(lib.clid_mapping_prep(None, 0, idx2.data_ptr(), iters, bs, 0, pool, None, 0, seed, counter + 1, None, 0.4, None,
                                     0, 0, 1, _lib.stream()), "clid_mapping_prep")
    g2 = idx2.cpu().numpy()
    assert (g2 != got).mean() > 0.99 and g2.max() < pool
    # spatially ordered variant: per iteration the SAME draws, in Morton order of the samples' voxels (stable)
    coords = (torch.rand((pool, 3), device=dev) * 80.0 - 40.0).contiguous()
    idx3 = torch.empty_like(idx)
    ws = torch.empty(int(lib.clid_mapping_prep_workspace_bytes(iters, bs)), device=dev, dtype=torch.uint8)
    _lib.check(lib.clid_mapping_prep(None, 0, idx3.data_ptr(), iters, bs, bs_new, pool, new_idx.data_ptr(), new_idx.shape[0], seed,
                                     counter, coords.data_ptr(), 0.4, ws.data_ptr(), 0, 0, 1, _lib.stream()), "clid_mapping_prep")
    g3 = idx3.cpu().numpy()
    assert np.array_equal(np.sort(g3, axis=1), np.sort(got, axis=1))  # a permutation of every iteration's draws

    def morton(c):
        out = np.zeros(c.shape[:-1], dtype=np.int64)
        for b in range(8):
            for a in range(3):
                out |= ((c[..., a] >> b) & 1) << (3 * b + a)
        return out

    cells = np.floor(coords.cpu().numpy()[g3] / np.float32(0.4)).astype(np.int64) & 255
    keys = morton(cells)
    assert (np.diff(keys, axis=1) >= 0).all() and len(np.unique(keys[0])) > 500
    # stable: equal keys keep their draw order
    pos_in_draw = {int(v): i for i, v in reversed(list(enumerate(got[0])))}
    same = np.nonzero(np.diff(keys[0]) == 0)[0]
    assert all(pos_in_draw[int(g3[0, i])] <= pos_in_draw[int(g3[0, i + 1])] for i in same[:200])

    # full 16 384-sample segments are shared by 8 blocks (splitters + per-range sort), the tail goes to one block: every
    # segment is a permutation of its draws, ordered by (Morton code, draw position); clustered coordinates give long ties
    iters2, bs2 = 2, 2 * 16384 + 1000
    coords2 = coords.clone()
    coords2[: pool // 2] = torch.floor(coords2[: pool // 2] / 3.2) * 3.2 + 0.1  # half of the pool shares a few hundred voxels
    raw = torch.empty((iters2, bs2), device=dev, dtype=torch.int64)
    _lib.check(lib.clid_mapping_prep(None, 0, raw.data_ptr(), iters2, bs2, 0, pool, None, 0, seed, 77, None, 0.4, None,
                                     0, 0, 1, _lib.stream()), "clid_mapping_prep")
    srt = torch.empty_like(raw)
    ws2 = torch.empty(int(lib.clid_mapping_prep_workspace_bytes(iters2, bs2)), device=dev, dtype=torch.uint8)
    _lib.check(lib.clid_mapping_prep(None, 0, srt.data_ptr(), iters2, bs2, 0, pool, None, 0, seed, 77, coords2.data_ptr(), 0.4,
                                     ws2.data_ptr(), 0, 0, 1, _lib.stream()), "clid_mapping_prep")
    raw_n, srt_n, c2 = raw.cpu().numpy(), srt.cpu().numpy(), coords2.cpu().numpy()
    for it in range(iters2):
        for lo in range(0, bs2, 16384):
            a, b = raw_n[it, lo:lo + 16384], srt_n[it, lo:lo + 16384]
            ka = morton(np.floor(c2[a] / np.float32(0.4)).astype(np.int64) & 255)
            want = a[np.argsort(ka, kind="stable")]  # stable: ties keep the draw order
            assert np.array_equal(b, want), (it, lo, int((b != want).sum()))

    # a rank's column window (a shard of a data-parallel run): the same values as the full call on the window -- widened to
    # whole 16 384-sample segments when ordering --, nothing written outside it
    for c0, nc, w0, w1 in ((16384, 16384, 16384, 32768), (20000, 5000, 16384, 32768), (30000, bs2 - 30000, 16384, bs2),
                           (0, 100, 0, 16384)):
        part = torch.full_like(raw, -7)
        _lib.check(lib.clid_mapping_prep(None, 0, part.data_ptr(), iters2, bs2, 0, pool, None, 0, seed, 77, coords2.data_ptr(),
                                         0.4, ws2.data_ptr(), c0, nc, 1, _lib.stream()), "clid_mapping_prep")
        pn = part.cpu().numpy()
        assert np.array_equal(pn[:, w0:w1], srt_n[:, w0:w1]), (c0, nc)
        assert (pn[:, :w0] == -7).all() and (pn[:, w1:] == -7).all(), (c0, nc)
    part = torch.full_like(raw, -7)
    _lib.check(lib.clid_mapping_prep(None, 0, part.data_ptr(), iters2, bs2, 0, pool, None, 0, seed, 77, None, 0.4, None, 777, 4321,
                                     1, _lib.stream()), "clid_mapping_prep")
    pn = part.cpu().numpy()
    assert np.array_equal(pn[:, 777:777 + 4321], raw_n[:, 777:777 + 4321]) and (pn[:, :777] == -7).all() and (pn[:, 777 + 4321:] == -7).all()

    # degenerate scene (60 % of the pool in ONE voxel): a key range then exceeds a sorting block's capacity and keeps its
    # draw order -- the batch is still a permutation of the draws, and the same one every time
    coords3 = coords.clone()
    coords3[: int(pool * 0.6)] = torch.tensor([1.0, 2.0, 3.0], device=dev)
    outs = []
    for _ in range(2):
        o3 = torch.empty_like(raw)
        _lib.check(lib.clid_mapping_prep(None, 0, o3.data_ptr(), iters2, bs2, 0, pool, None, 0, seed, 77, coords3.data_ptr(), 0.4,
                                         ws2.data_ptr(), 0, 0, 1, _lib.stream()), "clid_mapping_prep")
        outs.append(o3.cpu().numpy())
    assert np.array_equal(outs[0], outs[1])
    for it in range(iters2):
        for lo in range(0, bs2, 16384):
            assert np.array_equal(np.sort(outs[0][it, lo:lo + 16384]), np.sort(raw_n[it, lo:lo + 16384]))

    # class-preserving order (decimation d > 1, the numerical eikonal term's stride): the positions col % d == 0 of every batch
    # -- its eikonal subset coord[::d], utils/mapper.py:700-704 -- hold exactly the draws of those positions, the other positions
    # the other draws, each class in (Morton code, draw position) order.  Host restatement, segment by segment; full segments
    # (shared by 8 blocks), the one-block tail, a shard's column window, and the overflow path (60 % of the pool in one voxel)
    def class_order(a, ka, col0, d):
        cols = col0 + np.arange(a.shape[0])
        cls = cols % d == 0
        out = np.empty_like(a)
        for c in (True, False):
            sel = np.nonzero(cls == c)[0]
            out[sel] = a[sel][np.argsort(ka[sel], kind="stable")]
        return out

    for d in (10, 7, 3):
        srt_d = torch.empty_like(raw)
        _lib.check(lib.clid_mapping_prep(None, 0, srt_d.data_ptr(), iters2, bs2, 0, pool, None, 0, seed, 77, coords2.data_ptr(), 0.4,
                                         ws2.data_ptr(), 0, 0, d, _lib.stream()), "clid_mapping_prep")
        sd = srt_d.cpu().numpy()
        for it in range(iters2):
            assert np.array_equal(np.sort(sd[it, ::d]), np.sort(raw_n[it, ::d]))  # the eikonal subset IS the draws' own [::d]
            for lo in range(0, bs2, 16384):
                a = raw_n[it, lo:lo + 16384]
                ka = morton(np.floor(c2[a] / np.float32(0.4)).astype(np.int64) & 255)
                want = class_order(a, ka, lo, d)
                assert np.array_equal(sd[it, lo:lo + 16384], want), (d, it, lo, int((sd[it, lo:lo + 16384] != want).sum()))
        part = torch.full_like(raw, -7)
        _lib.check(lib.clid_mapping_prep(None, 0, part.data_ptr(), iters2, bs2, 0, pool, None, 0, seed, 77, coords2.data_ptr(),
                                         0.4, ws2.data_ptr(), 20000, 5000, d, _lib.stream()), "clid_mapping_prep")
        pn = part.cpu().numpy()
        assert np.array_equal(pn[:, 16384:32768], sd[:, 16384:32768]) and (pn[:, :16384] == -7).all() and (pn[:, 32768:] == -7).all()
        o3 = torch.empty_like(raw)
        _lib.check(lib.clid_mapping_prep(None, 0, o3.data_ptr(), iters2, bs2, 0, pool, None, 0, seed, 77, coords3.data_ptr(), 0.4,
                                         ws2.data_ptr(), 0, 0, d, _lib.stream()), "clid_mapping_prep")
        o3n = o3.cpu().numpy()
        for it in range(iters2):
            assert np.array_equal(np.sort(o3n[it, ::d]), np.sort(raw_n[it, ::d]))
            for lo in range(0, bs2, 16384):
                assert np.array_equal(np.sort(o3n[it, lo:lo + 16384]), np.sort(raw_n[it, lo:lo + 16384]))
    # a batch size that is no multiple of 4: the key array of every second iteration starts off a 16-byte boundary (the ordering
    # blocks' scalar-load path), three iterations of one full segment + an odd tail
    it3, bs3 = 3, 16384 + 1001
    raw3, srt3 = torch.empty((it3, bs3), device=dev, dtype=torch.int64), torch.empty((it3, bs3), device=dev, dtype=torch.int64)
    ws3 = torch.empty(int(lib.clid_mapping_prep_workspace_bytes(it3, bs3)), device=dev, dtype=torch.uint8)
    _lib.check(lib.clid_mapping_prep(None, 0, raw3.data_ptr(), it3, bs3, 0, pool, None, 0, seed, 31, None, 0.4, None, 0, 0, 1,
                                     _lib.stream()), "clid_mapping_prep")
    _lib.check(lib.clid_mapping_prep(None, 0, srt3.data_ptr(), it3, bs3, 0, pool, None, 0, seed, 31, coords2.data_ptr(), 0.4,
                                     ws3.data_ptr(), 0, 0, 10, _lib.stream()), "clid_mapping_prep")
    r3, s3 = raw3.cpu().numpy(), srt3.cpu().numpy()
    for it in range(it3):
        for lo in range(0, bs3, 16384):
            a = r3[it, lo:lo + 16384]
            ka = morton(np.floor(c2[a] / np.float32(0.4)).astype(np.int64) & 255)
            assert np.array_equal(s3[it, lo:lo + 16384], class_order(a, ka, lo, 10)), (it, lo)
    # (16 blocks per segment: the shape of the per-frame calls)
    one = torch.empty((1, 16384), device=dev, dtype=torch.int64)
    one_raw = torch.empty_like(one)
    ws1 = torch.empty(int(lib.clid_mapping_prep_workspace_bytes(1, 16384)), device=dev, dtype=torch.uint8)
    _lib.check(lib.clid_mapping_prep(None, 0, one_raw.data_ptr(), 1, 16384, 0, pool, None, 0, seed, 5, None, 0.4, None, 0, 0, 1,
                                     _lib.stream()), "clid_mapping_prep")
    _lib.check(lib.clid_mapping_prep(None, 0, one.data_ptr(), 1, 16384, 0, pool, None, 0, seed, 5, coords2.data_ptr(), 0.4,
                                     ws1.data_ptr(), 0, 0, 10, _lib.stream()), "clid_mapping_prep")
    a = one_raw.cpu().numpy()[0]
    ka = morton(np.floor(c2[a] / np.float32(0.4)).astype(np.int64) & 255)
    assert np.array_equal(one.cpu().numpy()[0], class_order(a, ka, 0, 10))

    # through the Mapper: two calls draw different batches, a second Mapper with the same seed reproduces them
    import bench
    from clid_slam_amd import HotPathConfig

    draws = []
    for rep in range(2):
        cfg = HotPathConfig()
        cfg.device, cfg.bs = dev, 2048
        torch.manual_seed(1)
        nm, dec, mp, scene = bench.build_scene(cfg, dev)
        mp.mapping(2)
        a = mp._keep[1].clone()
        mp.mapping(2)
        b = mp._keep[1].clone()
        assert a.shape == (2, 2048) and int(a.max()) < mp.pool_sample_count and not torch.equal(a, b)
        assert torch.isfinite(mp.last_losses).all() and float(mp.last_losses[:, 0].min()) > 0
        draws.append((a, b))
    assert torch.equal(draws[0][0], draws[1][0]) and torch.equal(draws[0][1], draws[1][1])


def test_batch_ordering_does_not_change_the_losses(monkeypatch):
    """The spatial ordering of a batch (clid_mapping_prep) is a pure reordering of the work: the eikonal subset coord[::10] of the
    ordered batch holds the same draws as that of the draws in their own order (utils/mapper.py:700-704 on the reference's
    unordered batch), so both loss terms of the first iteration agree to summation order -- with a spatially stratified subset the
    eikonal mean differs in the second digit."""
    import bench
    from clid_slam_amd import HotPathConfig

    dev = "cuda:0"
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("CLID_SORT_BATCH", mode)
        cfg = HotPathConfig()
        cfg.device, cfg.bs = dev, 16384 + 4096  # (a full segment shared by several blocks + a tail)
        torch.manual_seed(1)
        nm, dec, mp, scene = bench.build_scene(cfg, dev)
        mp.mapping(3)
        torch.cuda.synchronize()
        out[mode] = (mp.last_losses.cpu().numpy().copy(), mp._keep[1].cpu().numpy().copy())
    (la, ia), (lb, ib) = out["1"], out["0"]
    assert not np.array_equal(ia, ib) and np.array_equal(np.sort(ia, axis=1), np.sort(ib, axis=1))  # same draws, another order
    assert np.array_equal(np.sort(ia[:, ::10], axis=1), np.sort(ib[:, ::10], axis=1))              # the same eikonal subsets
    assert abs(la[0, 1] - lb[0, 1]) <= 2e-6 and abs(la[0, 2] - lb[0, 2]) <= 5e-6, (la[0], lb[0])
    assert np.abs(la - lb).max() <= 2e-4, (la, lb)  # later iterations: eps = 1e-15 sign chaos of single Adam entries apart


@pytest.mark.parametrize("layer_norm", [False, True])
def test_mapping_with_internal_draws_replays_on_the_oracle(layer_norm, monkeypatch):
    """Mapper.mapping with its own (spatially ordered) batch draws: the batches it used are replayed on the CPU oracle
    from a snapshot of the state -- losses, features, decoder and certainties agree as in the teacher-forced tests."""
    import bench
    from clid_slam_amd import HotPathConfig
    from oracle import cpu_ref as O
    from test_sequence import _oracle_state

    monkeypatch.setenv("CLID_SORT_BATCH", "1")  # (auto orders only calls of >= 32 iterations)
    dev = "cuda:0"
    cfg = HotPathConfig()
    cfg.device, cfg.bs, cfg.layer_norm_on = dev, 4096, layer_norm
    cfg.buffer_size = 2_000_003
    nm, dec, mp, scene = bench.build_scene(cfg, dev)
    st = _oracle_state(nm, cfg)
    od = O.DecoderParams(*[p.detach().cpu().clone() for p in dec.flat_params()], sdf_scale=dec.sdf_scale)
    opool = O.SamplePool(mp.global_coord_pool.cpu().clone(), mp.sdf_label_pool.cpu().clone(), mp.time_pool.cpu().clone(),
                         mp.weight_pool.cpu().clone())
    mp.mapping(3)
    idx = mp._keep[1].cpu()
    assert idx.shape == (3, 4096)
    keys = torch.floor(mp.global_coord_pool.cpu()[idx[0]] / cfg.voxel_size_m)
    assert len(torch.unique(keys, dim=0)) > 100  # a real batch, and ordered: consecutive samples are neighbours
    # (the order runs within the two classes of positions -- the eikonal lattice [::10] and the others --, see clid_mapping_prep)
    lattice = torch.arange(keys.shape[0]) % cfg.gradient_decimation == 0
    for sub, bound in ((keys[~lattice], 2.0), (keys[lattice], 6.0)):
        step = (sub[1:] - sub[:-1]).abs().max(dim=1).values.float().median()
        assert float(step) <= bound, float(step)
    recs = O.mapping_iters(st, od, opool, idx, O.LoopConfig(sigma=mp.sdf_scale), record=True)
    got = mp.last_losses.cpu()
    for it, r in enumerate(recs):
        assert abs(float(got[it, 0]) - float(r["loss"])) <= 2e-5, (it, got[it], r["loss"])
    noise = torch.zeros_like(recs[-1]["theta"], dtype=torch.bool)
    for r in recs:
        ga = r["grad_theta"].abs()
        noise |= (ga < 1e-10) & (ga.max(dim=1, keepdim=True).values > 0)
    err = (nm.local_geo_features.detach().cpu() - recs[-1]["theta"]).abs()
    assert float(err[~noise].max()) <= 1e-4 and float(err.max()) <= cfg.lr * 3 * 1.01  # (only entries the oracle's mask names may differ)
    for t, o in zip(dec.flat_params(), recs[-1]["dec"]):
        assert float((t.detach().cpu() - o).abs().max()) <= 1e-4


def test_pickle_after_gpu_calls_carries_no_device_mirror(env, tmp_path):
    """utils/tools.py:347-367 pickles the whole NeuralPoints module (`pin_map.pth`).  After the GPU calls that build the
    mirrors -- mapping() (local probe table + cell directory, cached per slot) and a query of the global map (global table) --
    the pickle must hold the reference's arrays only: no probe table, no directory, no cached buffers, no event."""
    import io
    import pickle

    g = gio.load("g6_loop_numerical_train_ln0.npz")
    p = gio.load("pool.npz")
    idx = gio.T(g["index_seq"]).to(torch.int64)[:2]
    cfg = env.config(bs=int(idx.shape[1]))
    nm = env.neural_points(cfg, base=p)
    dec = env.decoder(cfg, g, "init_")
    mp, _ = env.mapper(cfg, nm, dec)
    mp.mapping(2, index_seq=idx.cuda())
    nm.radius_neighborhood_search(gio.T(gio.load("g1_search.npz")["x"]).cuda())
    nm.prefetch_local_table(torch.cuda.Stream())
    torch.cuda.synchronize()
    assert nm._tables and nm.__dict__.get("_table_bufs") and nm.__dict__.get("_cdir_bufs")
    st = nm.__getstate__()
    for k in ("_table_bufs", "_cdir_bufs", "_table_event", "_travel32_cache", "_stencils", "_gbuf"):
        assert k not in st, k
    assert st["_tables"] == {}
    # what the pickle may weigh: the tensors of the reference's attribute list (+ the stencil constants)
    def nbytes(o, seen):
        if isinstance(o, torch.Tensor):
            key = o.untyped_storage().data_ptr()
            if key in seen:
                return 0
            seen.add(key)
            return o.untyped_storage().nbytes()
        if isinstance(o, dict):
            return sum(nbytes(v, seen) for v in o.values())
        if isinstance(o, (list, tuple)):
            return sum(nbytes(v, seen) for v in o)
        return 0
    expected = nbytes(st, set())
    path = tmp_path / "m.pth"
    torch.save(nm, path)
    size = path.stat().st_size
    mirrors = nbytes(nm.__dict__.get("_table_bufs"), set()) + nbytes(nm.__dict__.get("_cdir_bufs"), set())
    assert mirrors > (1 << 20)                       # (they exist and are large: the test means something)
    assert size < expected + (1 << 16), (size, expected, mirrors)
    nm2 = torch.load(path, weights_only=False)
    assert nm2._tables == {} and "_table_bufs" not in nm2.__dict__
    # a restored map answers the same search (mirrors rebuilt on demand, directory walk included)
    x = gio.T(gio.load("g1_search.npz")["x"]).cuda()
    d2a, ia = nm.radius_neighborhood_search(x)
    d2b, ib = nm2.radius_neighborhood_search(x)
    assert torch.equal(ia, ib) and torch.equal(d2a, d2b)
