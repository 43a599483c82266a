"""GPU: the touched-row bookkeeping of the hoisted-search loop (include/clid_native.h clid_train_args.touch_ws).

The searches of a chunk flag every map row every iteration will touch; `clid_train_touch_scan` turns the flags into
per-iteration bit sets, prefix sums and counts.  Single GPU: `k_adam_all` visits only rows touched so far in the call.
Sharded: each iteration all-reduces [848 | 9 floats per row it touches].  Both must reproduce the dense path: same
reference loop (G6), rows the reference never touched stay bit-identical (utils/tools.py:205-255: a row with
g = m = v = 0 gets a zero update), and the flag sets equal the neighbour ids of the search records."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

import golden_io as gio
import test_hip_parity as T
from test_hip_parity import maxerr

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import shim_io

    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return shim_io


@pytest.fixture()
def sparse_on(monkeypatch):
    monkeypatch.setenv("CLID_SPARSE", "1")


@pytest.mark.parametrize("mode,frozen,ln", [("numerical", False, 0), ("numerical", False, 1), ("numerical", True, 0)])
def test_reference_loop_with_the_touched_row_sweep(env, sparse_on, mode, frozen, ln):
    """G6 (the reference's own loop outputs) with the touched-row Adam sweep forced on."""
    T.test_mapping_loop_g6(env, mode, frozen, ln)


def _run(env, iters, bs, sparse, seed=11, ln=False, frozen=False, variant=None):
    from clid_slam_amd.tools import freeze_model

    os.environ["CLID_SPARSE"] = "1" if sparse else "0"
    try:
        p = gio.load("pool.npz")
        g = gio.load("g6_loop_numerical_train_ln0.npz")
        cfg = env.config(bs=bs, layer_norm_on=ln)
        nm = env.neural_points(cfg, base=p)
        dec = env.decoder(cfg, g, "init_")
        if frozen:
            freeze_model(dec)
        mp, _ = env.mapper(cfg, nm, dec)
        if variant is not None:
            mp.decode_variant = variant
        gen = torch.Generator().manual_seed(seed)
        idx = torch.randint(0, p["coord"].shape[0], (iters, bs), generator=gen).cuda()
        mp.mapping(iters, index_seq=idx)
        torch.cuda.synchronize()
        return (nm.local_geo_features.detach().cpu().clone(), [t.detach().cpu().clone() for t in dec.flat_params()],
                nm.local_point_certainties.cpu().clone(), nm.local_point_ts_update.cpu().clone(), mp.last_losses.cpu().clone(), mp)
    finally:
        os.environ.pop("CLID_SPARSE", None)


@pytest.mark.parametrize("iters,bs,ln,frozen", [(40, 2048, False, False), (36, 1024, True, True), (3, 16384, False, False)])
def test_touched_row_sweep_equals_the_dense_sweep(env, iters, bs, ln, frozen):
    """More iterations than one chunk holds (32): the first-touch bookkeeping carries over chunk boundaries.  Small batches
    on the fixture map leave many rows untouched for many iterations -- the regime the sweep exists for."""
    a = _run(env, iters, bs, False, ln=ln, frozen=frozen)
    b = _run(env, iters, bs, True, ln=ln, frozen=frozen)
    assert float((a[4] - b[4]).abs().max()) <= 2e-6
    moved = (a[0] - gio.T(gio.load("pool.npz")["base_geo_features"])[gio.T(gio.load("state.npz")["local_mask"])]).abs().amax(1) > 0
    # rows the dense sweep never moved are bit-identical under the touched-row sweep too (and vice versa)
    assert torch.equal(a[0][~moved], b[0][~moved])
    d = (a[0] - b[0]).abs()
    # two runs of the SAME path differ by the order of their atomics; Adam's eps = 1e-15 amplifies sign flips of
    # cancellation residues to +-lr per iteration: bound the bulk tightly, the tail by lr * iters
    assert float(d.max()) <= 0.01 * iters + 1e-6
    assert float((d > 1e-4).float().mean()) <= 2e-3, float((d > 1e-4).float().mean())
    for x, y in zip(a[1], b[1]):
        assert maxerr(x, y) <= (1e-4 if not frozen else 0.0)
    assert maxerr(a[2], b[2]) <= 1e-3 and torch.equal(a[3], b[3])
    assert 0 < int(moved.sum()) < moved.numel()


def _check_tile_number_blocks(rec, n_it, n_tasks, ids, live):
    """The number blocks behind every iteration's records (k_tile_number): per tile (two consecutive tasks) the pairs
    (query, neighbour) numbered per distinct map row -- rowid[number of pair (q, k)] == id of the pair, the numbers are a
    bijection onto the tile's distinct ids, 255 marks pairs without a neighbour (or of a padding slot)."""
    n_tiles = (n_tasks + 1) // 2
    per = rec.numel() // n_it
    # (behind the number blocks: the iteration's deferred list of the directory search, train_common.hpp)
    blocks = rec.view(n_it, per)[:, n_tasks * 192:n_tasks * 192 + n_tiles * 128].contiguous().view(torch.int32).cpu().numpy().reshape(n_it, n_tiles, 128)
    for it in range(n_it):
        for tile in range(0, n_tiles, 7):
            b = blocks[it, tile]
            count = int(b[96])
            rows = b[:count]
            nums = b[100:124].view(np.uint8).reshape(16, 6)
            want_ids = set()
            for q in range(16):
                task = 2 * tile + q // 8
                for k in range(6):
                    j = int(ids[it, task, q % 8, k]) if task < n_tasks and live[it, task, q % 8] else -1
                    if j < 0:
                        assert nums[q, k] == 255, (it, tile, q, k)
                    else:
                        assert nums[q, k] < count and rows[nums[q, k]] == j, (it, tile, q, k)
                        want_ids.add(j)
            assert len(set(rows.tolist())) == count == len(want_ids) and set(rows.tolist()) == want_ids


def test_flag_sets_equal_the_neighbour_ids_of_the_records(env):
    """clid_train_search + clid_train_touch_scan against a host restatement: bit i of iteration t = some live query of
    iteration t has neighbour i; counts = popcounts; prefix = exclusive scan; the running union carries over chunks."""
    from clid_slam_amd import _lib

    lib = _lib.load()
    p = gio.load("pool.npz")
    g = gio.load("g6_loop_numerical_train_ln0.npz")
    bs, n_it = 1024, 5
    cfg = env.config(bs=bs)
    nm = env.neural_points(cfg, base=p)
    dec = env.decoder(cfg, g, "init_")
    mp, _ = env.mapper(cfg, nm, dec)
    view, keep = nm._map_view(True)
    M = int(view.M)
    gen = torch.Generator().manual_seed(2)
    idx = torch.randint(0, p["coord"].shape[0], (n_it, bs), generator=gen).cuda()
    ta = _lib.TrainArgs()
    ta.pool_coord, ta.pool_label = mp.global_coord_pool.data_ptr(), mp.sdf_label_pool.data_ptr()
    ta.pool_ts, ta.pool_weight = mp.time_pool.data_ptr(), mp.weight_pool.data_ptr()
    ta.bs, ta.decimation, ta.batch_offset, ta.eikonal_mode = bs, 10, 0, 1
    ta.fd_eps, ta.loss_weight_on = float(cfg.voxel_size_m * cfg.num_grad_step_ratio), 1
    ta.grad_stride, ta.decode_variant, ta.pipeline = _lib.GRAD_ROW16, 1, 1
    chunk = int(lib.clid_train_chunk_iters(C.byref(ta)))
    assert 1 <= chunk <= 32
    m_cap = M + 1000
    stride = int(lib.clid_touch_stride(m_cap))
    assert stride % 256 == 0 and stride >= m_cap + 1
    ws = torch.zeros(int(lib.clid_touch_workspace_bytes(m_cap, chunk)), device="cuda", dtype=torch.uint8)
    ta.touch_ws, ta.touch_stride = ws.data_ptr(), stride
    rec = torch.empty(int(lib.clid_train_search_floats(bs, 0, 10, 1, n_it)), device="cuda")
    W = stride // 32
    seen = np.zeros(stride, bool)
    for it0 in (0, n_it):  # two "chunks" of one call: the second continues the running union
        _lib.check(lib.clid_train_search(C.byref(view), C.byref(ta), n_it, idx.data_ptr(), bs, rec.data_ptr(), _lib.stream()),
                   "clid_train_search")
        torch.cuda.synchronize()
        flags = ws[: n_it * stride].view(n_it, stride).cpu().numpy().copy()
        r = env.task_records(rec, n_it, int(lib.clid_train_search_tasks(bs, 0, 10, 1))).cpu()
        live = r[:, :, 0:8, 3].contiguous().view(torch.int32) >= 0                        # [it, task, slot]
        ids = r[:, :, 16:48, :].reshape(n_it, -1, 8, 8, 2)[..., :6, 1].contiguous().view(torch.int32)  # [it, task, slot, k]
        want = np.zeros((n_it, stride), bool)
        for it in range(n_it):
            j = ids[it][live[it]].reshape(-1).numpy()
            want[it, j[j >= 0]] = True
        assert np.array_equal(flags.astype(bool), want)
        if it0 == 0:
            _check_tile_number_blocks(rec, n_it, int(lib.clid_train_search_tasks(bs, 0, 10, 1)), ids.numpy(), live.numpy())
        counts = (C.c_int32 * 32)()
        _lib.check(lib.clid_train_touch_scan(C.byref(ta), M, n_it, it0, counts, _lib.stream()), "clid_train_touch_scan")
        assert not ws[: chunk * stride].any()  # cleared for the next chunk
        words = ws[chunk * stride:].view(torch.int32).cpu().numpy().view(np.uint32)
        bits, cumb, wpre = (words[k * chunk * W:(k + 1) * chunk * W].reshape(chunk, W) for k in range(3))
        for it in range(n_it):
            wbits = np.packbits(want[it].reshape(W, 32), axis=1, bitorder="little").view(np.uint32).reshape(W)
            assert np.array_equal(bits[it], wbits)
            pc = np.array([bin(int(x)).count("1") for x in wbits])
            assert counts[it] == pc.sum() == want[it].sum()
            assert np.array_equal(wpre[it], np.concatenate(([0], np.cumsum(pc)[:-1])).astype(np.uint32))
            seen |= want[it]
            assert np.array_equal(cumb[it], np.packbits(seen.reshape(W, 32), axis=1, bitorder="little").view(np.uint32).reshape(W))


def test_two_mappers_interleaved_and_threads(env):
    """ABI 3 keeps no state between calls: two Mappers with DIFFERENT decode kernels and batch sizes interleaved on one
    host thread, and the decode / Adam halves of an iteration issued from different host threads, give the results of
    running each alone (ABI 2 handed the partial-row count from clid_train_decode to clid_train_adam through a
    thread_local, and kept the kernel choice in a process-global)."""
    import threading

    from clid_slam_amd import _lib

    lib = _lib.load()
    alone = {}
    for tag, bs, variant in (("a", 2048, 1), ("b", 1000, 0)):
        alone[tag] = _run(env, 3, bs, False, seed=5, variant=variant)

    p = gio.load("pool.npz")
    g = gio.load("g6_loop_numerical_train_ln0.npz")
    objs = {}
    for tag, bs, variant in (("a", 2048, 1), ("b", 1000, 0)):
        cfg = env.config(bs=bs)
        nm = env.neural_points(cfg, base=p)
        dec = env.decoder(cfg, g, "init_")
        mp, _ = env.mapper(cfg, nm, dec)
        mp.decode_variant = variant
        gen = torch.Generator().manual_seed(5)
        idx = torch.randint(0, p["coord"].shape[0], (3, bs), generator=gen).cuda()
        objs[tag] = (nm, dec, mp, idx)
    # interleave: one iteration of A, one of B, ... each a 1-iteration mapping() call would restart Adam, so drive the two
    # halves through the C ABI directly: decode(A), decode(B), adam(A) on a second thread, adam(B)
    state = {}
    for tag, (nm, dec, mp, idx) in objs.items():
        bs = idx.shape[1]
        view, keep = nm._map_view(True)
        n_rows = nm.local_geo_features.shape[0]
        grad = torch.zeros(_lib.GRAD_FEAT_OFFSET16 + n_rows * 16, device="cuda")
        m, v = torch.zeros(n_rows * 8, device="cuda"), torch.zeros(n_rows * 8, device="cuda")
        mm, vm = torch.zeros(848, device="cuda"), torch.zeros(848, device="cuda")
        losses = torch.zeros((3, 4), device="cuda")
        ws = torch.empty(int(lib.clid_train_workspace_floats(bs, 10, 1)), device="cuda")
        rec = torch.empty(int(lib.clid_train_search_floats(bs, 0, 10, 1, 3)), device="cuda")
        W1, b1, W2, b2 = dec.flat_params()
        ta = _lib.TrainArgs()
        ta.pool_coord, ta.pool_label = mp.global_coord_pool.data_ptr(), mp.sdf_label_pool.data_ptr()
        ta.pool_ts, ta.pool_weight = mp.time_pool.data_ptr(), mp.weight_pool.data_ptr()
        ta.bs, ta.decimation, ta.batch_offset, ta.eikonal_mode = bs, 10, 0, 1
        ta.fd_eps = float(mp.config.voxel_size_m * mp.config.num_grad_step_ratio)
        ta.inv_n_main, ta.inv_n_eik = 1.0 / bs, 1.0 / ((bs + 9) // 10)
        ta.sigma, ta.weight_e, ta.loss_weight_on, ta.train_decoder = float(mp.sdf_scale), 0.5, 1, 1
        ta.W1, ta.b1, ta.W2, ta.b2 = W1.data_ptr(), b1.data_ptr(), W2.data_ptr(), b2.data_ptr()
        ta.sdf_scale, ta.defer_reduce = float(dec.sdf_scale), 1
        ta.grad, ta.ws = grad.data_ptr(), ws.data_ptr()
        ta.grad_stride, ta.decode_variant, ta.pipeline = 16, mp.decode_variant, 1
        aa = _lib.AdamArgs()
        aa.feat, aa.grad, aa.m, aa.v = nm.local_geo_features.data_ptr(), grad.data_ptr(), m.data_ptr(), v.data_ptr()
        aa.W1, aa.b1, aa.W2, aa.b2, aa.m_mlp, aa.v_mlp = ta.W1, ta.b1, ta.W2, ta.b2, mm.data_ptr(), vm.data_ptr()
        aa.n_feat, aa.lr, aa.beta1, aa.beta2, aa.eps, aa.weight_decay = n_rows * 8, 0.01, 0.9, 0.99, 1e-15, 0.0
        aa.train_decoder, aa.grad_stride = 1, 16
        aa.cert, aa.n_cert = nm.local_point_certainties.data_ptr(), int(nm.local_point_certainties.shape[0])
        _lib.check(lib.clid_train_search(C.byref(view), C.byref(ta), 3, idx.data_ptr(), bs, rec.data_ptr(), _lib.stream()), "search")
        state[tag] = (view, keep, ta, aa, rec, losses, (grad, m, v, mm, vm, ws), idx)
    s = _lib.stream()
    for it in range(3):
        for tag in ("a", "b"):
            view, keep, ta, aa, rec, losses, _, idx = state[tag]
            ta.index, ta.loss_out = idx.data_ptr() + it * idx.shape[1] * 8, losses.data_ptr() + it * 16
            per = rec.numel() // 3
            _lib.check(lib.clid_train_decode(C.byref(view), C.byref(ta), rec.data_ptr() + it * per * 4, s), "decode")
        errs = []

        def adam(tag):
            view, keep, ta, aa, rec, losses, _, idx = state[tag]
            aa.step = it + 1
            torch.cuda.set_device(0)
            errs.append(lib.clid_train_adam(C.byref(aa), C.byref(ta), s))
        th = threading.Thread(target=adam, args=("a",))
        th.start()
        th.join()
        adam("b")
        assert errs == [0, 0]
    torch.cuda.synchronize()
    for tag in ("a", "b"):
        nm, dec, mp, idx = objs[tag]
        ref = alone[tag]
        assert float((state[tag][5].cpu() - ref[4]).abs().max()) <= 2e-6
        d = (nm.local_geo_features.detach().cpu() - ref[0]).abs()
        assert float((d > 1e-4).float().mean()) <= 2e-3 and float(d.max()) <= 0.031
        for x, y in zip(dec.flat_params(), ref[1]):
            assert maxerr(x, y) <= 1e-4


def test_bench_spawns_its_ranks(tmp_path):
    """`python bench.py --gpus 2` with no launcher around it starts 2 ranks itself (gloo: both on this one GPU) and reports
    n_gpus 2; with the RCCL backend and fewer devices than ranks it fails loudly instead of running one rank."""
    import json
    import subprocess
    import sys

    # (no rendezvous variables of a surrounding launcher, and none of the CLID_* switches other tests of this process set)
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT") and not k.startswith("CLID_")}
    cmd = [sys.executable, os.path.join(T.__file__.rsplit("/tests/", 1)[0], "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--frame-calls", "0", "--no-cpu-baseline", "--bs", "2048", "--exchange-ab", "all"]
    out = subprocess.run(cmd + ["--backend", "gloo"], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 4096 and line["config"]["bs_per_gpu"] == 2048
    # one run = the A/B of the gradient exchange: the headline on the default (collective, dense at this map size), the
    # compact payload over the collective, and (--exchange-ab all: opt-in since round 5, the transport has never run across
    # devices) the compact payload over the peer-mapped buffers, ms per step each
    legs = line["config"]["gradient_exchange"]
    assert set(legs) >= {"rccl_dense", "rccl_compact", "p2p_compact"}
    assert legs["rccl_dense"].get("headline") and abs(legs["rccl_dense"]["ms_per_step"] - line["ms_per_step"]) < 1e-9
    assert legs["rccl_dense"]["mode"] == "dense" and legs["rccl_compact"]["mode"] == "compact"
    assert legs["rccl_compact"]["transport"] != "peer-mapped" and legs["p2p_compact"]["transport"] == "peer-mapped"
    assert legs["rccl_compact"]["bytes_per_iter"] < legs["rccl_dense"]["bytes_per_iter"]
    assert all(legs[k]["ms_per_step"] > 0 and legs[k]["p2p_fallbacks"] == 0 for k in ("rccl_dense", "rccl_compact", "p2p_compact"))
    if torch.cuda.device_count() < 2:
        out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
        assert out.returncode != 0 and "GPU(s) visible" in (out.stderr + out.stdout)
