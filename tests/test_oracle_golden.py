"""Pin the CPU oracle (oracle/cpu_ref.py) against outputs of the reference itself (fixtures G1-G6,
SURVEY.md section 8c).  CPU only."""
import numpy as np
import pytest
import torch

import golden_io as gio
from oracle import cpu_ref as O


def close(a, b, tol=1e-6, what=""):
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    err = np.abs(a.astype(np.float64) - np.asarray(b, dtype=np.float64)).max() if a.size else 0.0
    assert err <= tol, f"{what}: max abs err {err:.3e} > {tol}"


def test_search_neighborhood_table():
    z = gio.load("state.npz")
    dx, mv = O.search_neighborhood(2, 0.5, 0.4)
    assert dx.shape == (81, 3)
    assert np.array_equal(dx.numpy(), z["neighbor_dx"])
    assert mv == float(gio.S(z["max_valid_dist2"]))


@pytest.mark.parametrize("tf", [False, True])
def test_g1_search(tf):
    g = gio.load("g1_search.npz")
    st = gio.map_state()
    d2, idx = O.radius_neighborhood_search(st, gio.T(g["x"]), time_filtering=tf)
    assert np.array_equal(idx.numpy().astype(np.int32), g[f"idx_tf{int(tf)}"])
    assert np.array_equal(d2.numpy(), g[f"dist2_tf{int(tf)}"])  # bit-exact
    # the fixture really contains invalid probes, collisions rejected by distance and time-filtered hits
    assert (idx >= 0).any() and (idx < 0).any()
    if tf:
        assert (g["idx_tf0"] >= 0).sum() > (g["idx_tf1"] >= 0).sum()


CASES = [(ln, wf, tm, loc) for ln in (0, 1) for wf in (1, 0) for tm in (1, 0) for loc in (1, 0) if loc or not (tm or ln)]


@pytest.mark.parametrize("ln,wf,tm,loc", CASES)
def test_g2_query(ln, wf, tm, loc):
    g = gio.load("g2_query.npz")
    st = gio.map_state(layer_norm_on=bool(ln), weighted_first=bool(wf))
    tag = f"ln{ln}_wf{wf}_tm{tm}_loc{loc}"
    x, ts = gio.T(g["x"]), gio.T(g["ts"])
    f, w, nn, cert, _ = O.query_feature(st, x, ts if loc else None, training_mode=bool(tm), query_locally=bool(loc))
    assert np.array_equal(nn.numpy().astype(np.int32), g["nn_" + tag])
    close(f, g["f_" + tag], 1e-6, "features")
    close(w, g["w_" + tag], 1e-6, "weights")
    close(cert, g["cert_" + tag], 1e-5, "certainty")
    if loc:
        close(st.local_point_certainties, g["post_cert_" + tag], 1e-4, "post certainties")
        assert np.array_equal(st.local_point_ts_update.numpy(), g["post_ts_" + tag])
    assert (nn.numpy() == 0).any() and (nn.numpy() > 6).any()


def test_g3_mlp():
    g = gio.load("g3_mlp.npz")
    dec = gio.decoder()
    close(O.mlp_sdf(dec, gio.T(g["f"])), g["sdf"], 1e-7, "sdf")


@pytest.mark.parametrize("ln", [0, 1])
def test_g4_gradient_autograd_and_closed_form(ln):
    g = gio.load("g4_grad.npz")
    st = gio.map_state(layer_norm_on=bool(ln))
    dec = gio.decoder()
    x = gio.T(g["x"]).clone().requires_grad_(True)
    f, _, nn, _, _ = O.query_feature(st, x, training_mode=False)[0:5]
    s = O.mlp_sdf(dec, f)
    gr = O.autograd_gradient(x, s)
    close(s, g[f"sdf_ln{ln}"], 1e-6, "sdf")
    close(gr, g[f"grad_ln{ln}"], 1e-5, "autograd gradient")
    s2, g2, nn2 = O.closed_form_sdf_and_gradient(st, dec, gio.T(g["x"]))
    close(s2, g[f"sdf_ln{ln}"], 1e-6, "closed-form sdf")
    close(g2, g[f"grad_ln{ln}"], 2e-5, "closed-form gradient")
    assert np.array_equal(nn2.numpy().astype(np.int32), g[f"nn_ln{ln}"])


def test_g5_loss():
    g = gio.load("g5_loss.npz")
    pred = gio.T(g["pred"]).clone().requires_grad_(True)
    gv = gio.T(g["g"]).clone().requires_grad_(True)
    l_bce = O.sdf_bce_loss(pred, gio.T(g["label"]), 0.055, gio.T(g["weight"]), True)
    l_eik = O.eikonal_loss(gv)
    (l_bce + 0.5 * l_eik).backward()
    close(l_bce, g["l_bce"], 1e-6)
    close(l_eik, g["l_eik"], 1e-6)
    close(pred.grad, g["dpred"], 1e-8)
    close(gv.grad, g["dg"], 1e-8)


def test_adam_matches_torch_optim():
    torch.manual_seed(0)
    p0 = torch.randn(50, 8)
    p_ref = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([p_ref], lr=0.01, betas=(0.9, 0.99), eps=1e-15)
    p = p0.clone()
    s = O.AdamState(torch.zeros_like(p), torch.zeros_like(p))
    for _ in range(4):
        g = torch.randn(50, 8)
        g[::3] = 0.0
        p_ref.grad = g.clone()
        opt.step()
        O.adam_step(p, g, s)
        assert torch.equal(p, p_ref.detach())


G6 = [("numerical", False, 0, "all"), ("numerical", False, 1, "all"), ("numerical", True, 0, "all"), ("analytic", False, 0, "all"),
      ("analytic", True, 0, "all"),
      # config.ekional_add_to (utils/mapper.py:779-789): the eikonal mean over the near-surface / the free-space decimated samples
      ("numerical", False, 0, "surface"), ("numerical", False, 0, "freespace"),
      # config.main_loss_type (utils/mapper.py:751-767) and Mapper.ba_done_flag (utils/mapper.py:646-658)
      ("numerical", False, 0, "all", "sdf_l1"), ("numerical", False, 0, "all", "sdf_l2"), ("numerical", False, 0, "all", "zhong"),
      ("numerical", False, 0, "all", "bce", True),
      # neuralpoints.weighted_first: False with the analytic eikonal term (utils/mapper.py:679-680, 695-696)
      ("analytic", False, 0, "all", "bce", False, False), ("analytic", False, 1, "all", "bce", False, False),
      # config.proj_correction_on (utils/mapper.py:57-69, 712-714): labels scaled by |cos(g, x - origin)|, g in the graph
      ("numerical", False, 0, "all", "bce", False, True, True), ("numerical", False, 1, "all", "bce", False, True, True),
      # config.consistency_loss_on (utils/mapper.py:716-741, 770-776), with the reference's recorded draws
      ("numerical", False, 0, "all", "bce", False, True, False, True), ("numerical", False, 1, "all", "bce", False, True, False, True)]


@pytest.mark.parametrize("case", G6, ids=lambda c: "-".join(str(x) for x in c))
def test_g6_mapping_loop(case):
    mode, frozen, ln, add_to, loss_type, ba, wf, proj, cons = tuple(case) + ("all", "bce", False, True, False, False)[len(case) - 3:]
    tag = (f"{mode}_{'frozen' if frozen else 'train'}_ln{ln}" + ("" if add_to == "all" else f"_eik{add_to}")
           + ("" if loss_type == "bce" else f"_{loss_type}") + ("_ba" if ba else "") + ("" if wf else "_wf0") + ("_proj" if proj else "")
           + ("_cons" if cons else ""))
    g = gio.load(f"g6_loop_{tag}.npz")
    st = gio.map_state(layer_norm_on=bool(ln), weighted_first=bool(wf))
    pool, praw = gio.sample_pool()
    if ba:  # the pool in the samples' sensor frames + the frames' poses; the world-frame pool is stale and must not be read
        pool.local_coord, pool.used_poses = gio.T(g["ba_coord_pool"]), gio.T(g["ba_used_poses"])
        pool.global_coord = pool.global_coord + 0.37
        moved = O.transform_batch(pool.local_coord, pool.used_poses[pool.time.long()])
        assert 0 < float((moved - gio.T(praw["coord"])).abs().max()) < 1e-5  # (the poses map the sensor frames back onto the scene)
    if proj:
        pool.frame_poses = gio.T(g["proj_used_poses"])
    dec = gio.decoder(g, "init_")
    if loss_type != "bce":
        dec.sdf_scale = 1.0  # model/decoder.py:51-53: the decoder's output scale is the logistic sigma only for the BCE loss
    lc = O.LoopConfig(numerical_grad=(mode == "numerical"), gradient_decimation=10 if mode == "numerical" else 1,
                      train_decoder=not frozen, ekional_add_to=add_to, main_loss_type=loss_type, proj_correction_on=bool(proj),
                      consistency_loss_on=bool(cons), weight_c=float(g["cons_weight_c"]) if cons else 0.5)
    cons_seq = None
    if cons:
        cons_seq = [(gio.T(g["cons_near_index"][it]).to(torch.int64), gio.T(g["cons_shift"][it])) for it in range(g["cons_shift"].shape[0])]
    index_seq = gio.T(g["index_seq"]).to(torch.int64)
    # a1: the batch composition rule reproduces the reference's batch from its recorded draws
    idx0 = torch.cat((gio.T(g["draw_hist0"]), gio.T(g["new_idx"])[gio.T(g["draw_pick0"])]))
    assert torch.equal(idx0, index_seq[0])
    recs = O.mapping_iters(st, dec, pool, index_seq, lc, record=True, consistency_seq=cons_seq)
    for it, r in enumerate(recs):
        close(r["sdf_loss"], g["loss_bce"][it], 2e-6, f"it{it} bce")
        close(r["loss"], g["loss_total"][it], 2e-6, f"it{it} total")
        rows = g[f"it{it}_grad_theta_rows"].astype(np.int64)
        gt = r["grad_theta"].numpy()
        dense = np.zeros_like(gt)
        dense[rows] = g[f"it{it}_grad_theta_vals"]
        scale = max(np.abs(dense).max(), 1e-12)
        assert np.abs(gt - dense).max() / scale < 1e-4, f"it{it} grad_theta"
        # rows with exactly zero gradient in the reference are exactly zero here
        untouched = np.ones(gt.shape[0], bool)
        untouched[rows] = False
        assert not gt[untouched].any()
        if not frozen:
            for n in ("W1", "b1", "W2", "b2"):
                ref_g = g[f"it{it}_grad_{n}"]
                sc = max(np.abs(ref_g).max(), 1e-12)
                assert np.abs(r["grad_" + n].numpy() - ref_g).max() / sc < 1e-4, f"it{it} grad_{n}"
        close(r["theta"], g[f"it{it}_theta"], 1e-4, f"it{it} theta")
        for n, t in zip(("W1", "b1", "W2", "b2"), r["dec"]):
            close(t, g[f"it{it}_{n}"], 1e-4, f"it{it} {n}")
        close(r["certainties"], g[f"it{it}_certainties"], 1e-3, f"it{it} certainties")
        assert np.array_equal(r["ts_update"].numpy(), g[f"it{it}_ts_update"])
    last = len(recs) - 1
    close(recs[last]["adam_m_theta"], g[f"it{last}_adam_m_theta"], 1e-6)
    close(recs[last]["adam_v_theta"], g[f"it{last}_adam_v_theta"], 1e-6)
    # a10: write-back
    z = gio.load("state.npz")
    st.geo_features = gio.T(praw["base_geo_features"]).clone()
    st.point_certainties = gio.T(praw["base_point_certainties"]).clone()
    tsu = gio.T(praw["base_point_ts_update"]).clone()
    O.assign_local_to_global(st, gio.T(g["local_mask"]), tsu)
    base = praw["base_geo_features"].copy()
    base[g["final_geo_rows"].astype(np.int64)] = g["final_geo_vals"]
    close(st.geo_features, base, 1e-4, "final geo features")
    close(st.point_certainties, g["final_point_certainties"], 1e-3)
    assert np.array_equal(tsu.numpy(), g["final_point_ts_update"])


@pytest.mark.parametrize("ln,wf", [(0, 1), (1, 1), (0, 0), (1, 0)])
def test_g8_tracking_measurement_model(ln, wf):
    """Row N1: the oracle's h_model against the reference's own IEKFOM.h_model output, for both `weighted_first` settings
    (wf = 0: every neighbour decoded, SDFs blended, the std-of-SDFs mask of utils/error_state_iekf.py:217-241 active)."""
    g = gio.load("g8_tracking.npz")
    st = gio.map_state(layer_norm_on=bool(ln), weighted_first=bool(wf))
    dec = gio.decoder()
    lo, hi = g["grad_window"]
    tag = f"ln{ln}" + ("" if wf else "_wf0")
    std_max = 0.25 if wf else 0.25 * float(g["max_sdf_std_ratio_wf0"][ln])
    z, H, vp, r_inv, valid = O.h_model(st, dec, gio.T(g["rot"]), gio.T(g["pos"]), gio.T(g["pc_imu"]),
                                       min_grad_norm=float(lo), max_grad_norm=float(hi), max_sdf_std=std_max)
    assert H.shape[0] == g[f"H6_{tag}"].shape[0] > 100
    close(z, g[f"z_{tag}"], 1e-6, "residual")
    close(H[:, :6], g[f"H6_{tag}"], 2e-5, "Jacobian")
    assert float(H[:, 6:].abs().max()) == 0.0
    close(vp, g[f"valid_points_{tag}"], 1e-5, "valid points")
    close(r_inv, g[f"R_inv_{tag}"], 1e-2, "R_inv")  # values ~1e3


def test_oracle_query_composition_vs_reference_mesher_g12():
    """Row N3: the oracle's query_feature -> decoder composition against the REFERENCE's own Mesher.query_points
    (fixture G12: global + local map, with and without layer norm, nn >= 1 / nn >= 4 masks)."""
    g = gio.load("g12_mesher.npz")
    x = gio.T(g["x"])
    for wf in (1, 0):
        for ln in (0, 1):
            for loc in (0, 1):
                st = gio.map_state(layer_norm_on=bool(ln), weighted_first=bool(wf))
                f, w, nn, _, _ = O.query_feature(st, x, training_mode=False, query_locally=bool(loc))
                sdf = O.mlp_sdf(gio.decoder(), f)
                if not wf:  # utils/mesher.py:130-138
                    sdf = (sdf * w).sum(dim=1).squeeze(1)
                sdf = torch.where(nn >= 1, sdf, torch.zeros(()))
                tag = f"ln{ln}_loc{loc}" + ("" if wf else "_wf0")
                assert float((sdf - gio.T(g[f"sdf_{tag}"])).abs().max()) <= 1e-6, tag
                assert torch.equal((nn >= 4), gio.T(g[f"mask_{tag}"]).bool()), tag


@pytest.mark.parametrize("ln", [0, 1])
def test_relu_ambiguous_rows_is_a_pure_checker_aid(ln):
    """`relu_ambiguous_rows` (the rows whose gradient may legitimately jump between two fp32 evaluations: gathered by a query
    with a decoder pre-activation on the ReLU kink) leaves the state untouched, is empty for tau = 0, grows with tau, only
    names rows the iteration really gathers, and recording it does not change the loop's records."""
    g = gio.load(f"g6_loop_numerical_train_ln{ln}.npz")
    st = gio.map_state(layer_norm_on=bool(ln))
    pool, _ = gio.sample_pool()
    dec = gio.decoder(g, "init_")
    lc = O.LoopConfig()
    index_seq = gio.T(g["index_seq"]).to(torch.int64)
    cert0, ts0 = st.local_point_certainties.clone(), st.local_point_ts_update.clone()
    r0, q0 = O.relu_ambiguous_rows(st, dec, pool, index_seq[0], lc, 0.0)
    r1, q1 = O.relu_ambiguous_rows(st, dec, pool, index_seq[0], lc, 1e-4)
    r2, q2 = O.relu_ambiguous_rows(st, dec, pool, index_seq[0], lc, 1e-2)
    assert torch.equal(st.local_point_certainties, cert0) and torch.equal(st.local_point_ts_update, ts0)
    assert r0.numel() == 0 and q0 == 0 and q1 <= q2 and q2 > 0
    assert set(r1.tolist()) <= set(r2.tolist())
    recs = O.mapping_iters(st, dec, pool, index_seq[:2], lc, record=True, ambiguity_tau=1e-2)
    touched = torch.nonzero((recs[0]["grad_theta"] != 0).any(1)).flatten()
    assert set(recs[0]["ambiguous_rows"].tolist()) <= set(touched.tolist()) | {int(st.local_geo_features.shape[0]) - 1}
    assert torch.equal(recs[0]["ambiguous_rows"], r2)
    close(recs[1]["loss"], g["loss_total"][1], 2e-6, "records unchanged by the ambiguity pass")


@pytest.mark.parametrize("ln", [0, 1])
@pytest.mark.parametrize("tau", [1e-5, 3e-5])
def test_relu_kink_row_bound_holds_for_forced_gates(ln, tau, monkeypatch):
    """The per-row bound `relu_ambiguous_rows(..., with_slack=True)` hands to the GPU checks (a listed row may differ from the
    oracle's gradient by at most strict tolerance + 1.25 x its bound; every other row is strict): checked oracle against
    oracle.  The decoder's ReLU gates of every (query, unit) inside the tau band are FORCED open, forced closed, or flipped at
    random -- the evaluations two correct fp32 implementations may disagree on -- and the gradient of the iteration
    (utils/mapper.py:642-836) is recomputed: rows outside the list do not move (<= 1e-6 of the largest entry), listed rows
    move by less than their bound, and rows nobody gathers stay exactly zero."""
    import torch.nn.functional as F

    g = gio.load(f"g6_loop_numerical_train_ln{ln}.npz")
    pool, _ = gio.sample_pool()
    dec = gio.decoder(g, "init_")
    lc = O.LoopConfig()
    idx = gio.T(g["index_seq"]).to(torch.int64)[0]
    st = gio.map_state(layer_norm_on=bool(ln))
    rows, nq, slack, gathered, dec_slack = O.relu_ambiguous_rows(st, dec, pool, idx, lc, tau, with_slack=True)
    assert nq > 0 and slack is not None and float(slack[rows].min()) >= 0.0 and float(slack.sum()) > 0.0
    assert dec_slack.shape[0] == dec.W1.numel() + 2 * dec.W1.shape[0] + 1 and float(dec_slack.min()) >= 0.0
    kink = torch.zeros(slack.shape[0], dtype=torch.bool)
    kink[rows] = True
    assert float(slack[~kink].abs().max()) == 0.0 and bool(gathered[rows].all())
    base_all = O.loss_and_grads(st, dec, pool, idx, lc)
    base = base_all["grad_theta"]
    base_dec = torch.cat([base_all["grad_" + n].reshape(-1) for n in ("W1", "b1", "W2", "b2")])
    assert not bool((base != 0).any(1)[~gathered].any())
    gmax, dmax = float(base.abs().max()), float(base_dec.abs().max())
    moved = moved_dec = 0.0
    for mode in (True, False, None):
        gen = torch.Generator().manual_seed(7)

        def forced(d, f, mode=mode):
            pre = F.linear(f, d.W1, d.b1)
            amb = pre.detach().abs() < tau
            m = torch.full_like(amb, bool(mode)) if mode is not None else (torch.rand(amb.shape, generator=gen) < 0.5)
            gate = torch.where(amb, m, pre.detach() > 0).to(pre.dtype)
            return F.linear(pre * gate, d.W2, d.b2).squeeze(1) * d.sdf_scale

        monkeypatch.setattr(O, "mlp_sdf", forced)
        other_all = O.loss_and_grads(gio.map_state(layer_norm_on=bool(ln)), dec, pool, idx, lc)
        monkeypatch.undo()
        other = other_all["grad_theta"]
        dd = (torch.cat([other_all["grad_" + n].reshape(-1) for n in ("W1", "b1", "W2", "b2")]) - base_dec).abs()
        # the decoder's gradient: entries of units nobody holds on the kink do not move, the others by less than their bound
        assert float(dd[dec_slack == 0].max()) <= 1e-6 * dmax
        assert float((dd - 1.25 * dec_slack).max()) <= 1e-5 * dmax
        moved_dec = max(moved_dec, float(dd.max()) / dmax)
        d = (other - base).abs().max(1).values
        assert float(d[~kink].max()) <= 1e-6 * gmax
        assert float((d - 1.25 * slack).max()) <= 1e-5 * gmax
        assert not bool((other != 0).any(1)[~gathered].any())
        moved = max(moved, float(d[kink].max()) / gmax)
    assert moved > 1e-4  # the forced gates really moved listed rows beyond the strict bar (the bound is doing work)
    assert moved_dec > 1e-5


@pytest.mark.skipif(not __import__("os").path.isdir("/root/reference"), reason="needs the reference tree (build container only)")
@pytest.mark.parametrize("seed", [101, 977])
def test_oracle_matches_the_reference_out_of_fixture(seed, tmp_path):
    """Beyond the committed vectors: the reference itself is run HERE on another scene with other draws (every seed of
    oracle/make_golden.py's G1-G6 generation shifted by `seed`: map, features, queries, decoder, batches, new-sample picks,
    consistency draws, BA poses) and the oracle's G1-G6 tests -- search bit-exact, query, mlp, gradient, loss, and EVERY branch of
    the mapping loop (17 G6 cases) -- run against that directory instead of tests/golden.  An oracle that had been fitted to the
    committed fixtures rather than to utils/mapper.py:642-836 would fail here.  (The GPU box has no reference: skipped there.)"""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, OMP_NUM_THREADS="1")
    out = str(tmp_path / "fresh")
    r = subprocess.run([sys.executable, os.path.join(root, "oracle", "make_golden.py"), "--fresh", out, "--seed", str(seed)],
                       capture_output=True, text=True, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    committed = np.load(os.path.join(root, "tests", "golden", "g6_loop_numerical_train_ln0.npz"))
    fresh = np.load(os.path.join(out, "g6_loop_numerical_train_ln0.npz"))
    assert not np.array_equal(committed["index_seq"], fresh["index_seq"])  # really other draws
    assert not np.array_equal(np.load(os.path.join(root, "tests", "golden", "state.npz"))["geo_features"],
                              np.load(os.path.join(out, "state.npz"))["geo_features"])  # ... on another map
    env["CLID_GOLDEN_DIR"] = out
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_oracle_golden.py"), "-q", "-x", "-p", "no:cacheprovider",
                        "-k", "test_search_neighborhood_table or test_g1 or test_g2 or test_g3 or test_g4 or test_g5 or test_g6 or test_g8 or g12"],
                       capture_output=True, text=True, env=env, cwd=root)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-1000:])
    assert " passed" in r.stdout and "failed" not in r.stdout
