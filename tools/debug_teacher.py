"""developer tool (GPU box): locate the entries where the teacher-forced gradient of config 5 differs from the oracle, and
compare the SDF of EVERY query slot of that iteration (HIP: clid_train_args.sdf_dbg) with the oracle's."""
import sys, os, json, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
import bench_sequence as BS
from oracle import cpu_ref as O
import clid_slam_amd  # noqa
from clid_slam_amd.mapper import Mapper

Mapper._probe_sdf = True
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 1
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 5e-5
orig = BS._check_against_oracle


def spy(snap, idx, nm, dec, mp, cfg, fid):
    st, od, pool, frozen, probes, recs = snap
    for t, (p, r) in enumerate(zip(probes, recs)):
        g0, gh = r["grad_theta"], p["theta"]
        gmax = float(g0.abs().max())
        amb = set(r.get("ambiguous_rows", torch.empty(0)).tolist())
        dd = (gh - g0).abs().max(1).values
        rows = [x for x in torch.nonzero(dd > thr * gmax).flatten().tolist() if x not in amb]
        if not rows:
            continue
        print(f"frame {fid} iter {t}: bad non-kink rows {rows[:8]} rel {float(dd[rows].max()) / gmax:.3e} gmax {gmax:.3e}")
        if t == 0:
            print("  (iteration 0: the snapshot's parameters were advanced in place by the oracle replay; per-slot comparison skipped)")
            continue
        st2 = copy.deepcopy(st)
        od2 = od
        if t > 0:
            st2.local_geo_features = recs[t - 1]["theta"].clone()
            od2 = O.DecoderParams(*[x.clone() for x in recs[t - 1]["dec"]], sdf_scale=od.sdf_scale)
        index = idx.cpu()[t]
        coord = pool.global_coord[index]
        d = cfg.gradient_decimation
        eps = cfg.voxel_size_m * cfg.num_grad_step_ratio
        x = coord[::d]
        pts = [coord]
        for a in range(3):
            e = torch.zeros(3)
            e[a] = eps
            pts += [x + e, x - e]
        allp = torch.cat(pts)
        with torch.no_grad():
            f, w, nn, _, qidx = O.query_feature(st2, allp, None, training_mode=False)
            pre = F.linear(f, od2.W1, od2.b1)
            sdf = O.mlp_sdf(od2, f)
        n, m = coord.shape[0], x.shape[0]
        rh, sh = p["records"], p["sdf_slots"]
        pos = rh[:, 8:16, 0].contiguous().view(torch.int32).reshape(-1).long()
        code = rh[:, 8:16, 1].contiguous().view(torch.int32).reshape(-1).long()
        live = pos >= 0
        # oracle query number of (batch position, code): code -1 = the sample, 2*axis + (sign > 0) = shifted copy
        a6 = torch.where(code >= 0, 2 * (code // 2) + (1 - code % 2), torch.zeros_like(code))
        oq = torch.where(code < 0, pos, n + a6 * m + pos // d)
        dsdf = torch.where(live, (sh.reshape(-1) - sdf[oq.clamp(min=0)]).abs(), torch.zeros(()))
        top = torch.topk(dsdf, 6)
        print("  max |sdf_hip - sdf_oracle| over all query slots:", [f"{v:.3e}" for v in top.values.tolist()])
        win = rh[:, 16:48].reshape(-1, 8, 4, 2).reshape(-1, 8, 2)  # per slot: 8 float2
        for sl in top.indices[:3].tolist():
            q = int(oq[sl])
            ids_h = win[sl, :6, 1].contiguous().view(torch.int32).tolist()
            print(f"   slot {sl} pos {int(pos[sl])} code {int(code[sl])} oracle q {q}: hip sdf {float(sh.reshape(-1)[sl]):.6f} oracle {float(sdf[q]):.6f} "
                  f"nn {int(nn[q])} min|pre| {float(pre[q].abs().min()):.3e}\n      hip w {[round(v, 6) for v in win[sl, :6, 0].tolist()]} ids {ids_h}\n"
                  f"      ora w {[round(float(v), 6) for v in w[q, :, 0]]} ids {qidx[q].tolist()}")
        for rr in rows[:2]:
            qs = torch.nonzero((qidx == rr).any(1)).flatten()
            th = st2.local_geo_features[rr]
            print(f"  row {rr}: gathered by {qs.numel()} query points; theta {th.tolist()} var {float(th.var(unbiased=False)):.3e}")
            print("     oracle g", g0[rr].tolist(), "\n        hip g", gh[rr].tolist())
            for q in qs[:10].tolist():
                k = int(torch.nonzero(qidx[q] == rr)[0])
                sl = torch.nonzero(live & (oq == q)).flatten()
                hs = float(sh.reshape(-1)[sl[0]]) if sl.numel() else float("nan")
                print(f"     q {q} nn {int(nn[q])} w {float(w[q, k, 0]):.5f} min|pre| {float(pre[q].abs().min()):.3e} sdf oracle {float(sdf[q]):.6f} hip {hs:.6f}")
    return orig(snap, idx, nm, dec, mp, cfg, fid)


BS._check_against_oracle = spy
cfg, rows, checks, _ = BS.run(frames, "cuda:0", check_frames=frames, quiet=True, teacher_iters=12)
for c in checks:
    print("frame", c["frame"], "strict x1e-4:", [round(r["dgrad_theta_rel"] * 1e4, 3) for r in c["teacher_forced"]],
          "decoder x1e-4:", [round(r.get("dgrad_decoder_rel", 0) * 1e4, 3) for r in c["teacher_forced"]])
