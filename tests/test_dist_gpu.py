"""GPU, world size 2 on ONE device: the sharded `Mapper.mapping` path end to end (batch slices with
`batch_offset`, global loss normalisers, `k_reduce_partials`, gradient all-reduce, side-effect merge) must
reproduce the single-process result.  Both ranks use cuda:0 and the gloo backend (the box has one GPU; on
a real node the same code runs one rank per GPU over RCCL)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(rank, world, port, out_dir, mode, sparse="0", p2p="1", every="0", fail_rank=-1):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import golden_io as gio
    import shim_io

    os.environ["CLID_P2P"] = p2p        # "1": the compact payload over peer-mapped buffers (csrc/p2p.hip) when they can be set up
    os.environ["CLID_TOUCH_ALL"] = every  # "1": every row on every iteration's list (no flag exchange / read-back per chunk)
    os.environ["CLID_SPARSE"] = sparse  # "1": compact exchange [848 | 9 floats per touched row] + touched-row Adam sweep
    if world > 1:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    p = gio.load("pool.npz")
    g = gio.load("g6_loop_numerical_train_ln0.npz")
    bs, iters = 4096, 3
    cfg = shim_io.config(bs=bs)
    if mode in ("analytic", "wf0_analytic"):
        cfg.numerical_grad, cfg.gradient_decimation = False, 1
    if mode in ("wf0", "wf0_analytic"):  # every neighbour decoded, SDFs blended (csrc/train_wf0.hip): shards like the default iteration
        cfg.weighted_first = False
    nm = shim_io.neural_points(cfg, base=p)
    dec = shim_io.decoder(cfg, g, "init_")
    mpr, _ = shim_io.mapper(cfg, nm, dec)
    gen = torch.Generator().manual_seed(21)
    idx = torch.randint(0, p["coord"].shape[0], (iters, bs), generator=gen).cuda()
    if fail_rank >= 0:
        # the peer-mapped exchange object is set up (collectively), then ONE rank's error word is raised the way a flag
        # wait that gave up raises it: every rank must notice, restore its state and repeat the call over the fallback
        from clid_slam_amd import _lib

        obj = _lib.p2p_exchange(dist, 1 << 20)
        assert obj is not None
        if rank == fail_rank:
            _lib.check(_lib.load().clid_debug_p2p_fail(obj, _lib.stream()), "clid_debug_p2p_fail")
    mpr.mapping(iters, index_seq=idx)
    torch.cuda.synchronize()
    if fail_rank >= 0:
        assert mpr.p2p_fallbacks == 1 and mpr.last_exchange["transport"] != "peer-mapped", (rank, mpr.p2p_fallbacks, mpr.last_exchange)
        assert _lib._p2p is False  # the transport stays ruled out for this process
    if rank == 0:
        np.savez(os.path.join(out_dir, f"w{world}.npz"), theta=nm.local_geo_features.detach().cpu().numpy(),
                 W1=dec.flat_params()[0].detach().cpu().numpy(), b2=dec.flat_params()[3].detach().cpu().numpy(),
                 cert=nm.local_point_certainties.cpu().numpy(), ts=nm.local_point_ts_update.cpu().numpy(),
                 loss=mpr.last_losses.cpu().numpy(),
                 exchange=np.array([0 if mpr.last_exchange is None else mpr.last_exchange["floats"],
                                    0 if mpr.last_exchange is None else int(mpr.last_exchange["mode"] == "compact"),
                                    0 if mpr.last_exchange is None else int(mpr.last_exchange["transport"] == "peer-mapped"),
                                    0 if mpr.last_exchange is None else int(mpr.last_exchange["every_row"])]))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("mode,sparse,p2p,every", [("numerical", "0", "1", "0"), ("analytic", "0", "1", "0"), ("wf0", "0", "0", "0"), ("wf0_analytic", "0", "0", "0"),
                                                   ("numerical", "1", "0", "0"),
                                                   ("numerical", "1", "1", "0"), ("numerical", "1", "1", "1"), ("numerical", "1", "0", "1")])
def test_two_ranks_equal_one(tmp_path, mode, sparse, p2p, every, monkeypatch):
    for k in ("CLID_P2P", "CLID_TOUCH_ALL", "CLID_SPARSE"):  # (_run sets them in THIS process for the one-rank run: restored afterwards)
        monkeypatch.setenv(k, os.environ.get(k, "0"))
    port = 29700 + (os.getpid() % 1000) + {"numerical": 0, "analytic": 1, "wf0": 16, "wf0_analytic": 17}[mode] + 2 * int(sparse) + 4 * int(p2p) + 8 * int(every)
    _run(0, 1, port, str(tmp_path), mode)
    os.environ.pop("CLID_SPARSE", None)
    mp.spawn(_run, args=(2, port, str(tmp_path), mode, sparse, p2p, every), nprocs=2, join=True)
    a = np.load(os.path.join(tmp_path, "w1.npz"))
    b = np.load(os.path.join(tmp_path, "w2.npz"))
    assert np.abs(a["loss"] - b["loss"]).max() <= 2e-6
    assert np.abs(a["theta"] - b["theta"]).max() <= 2e-5
    assert np.abs(a["W1"] - b["W1"]).max() <= 2e-5 and np.abs(a["b2"] - b["b2"]).max() <= 2e-5
    assert np.abs(a["cert"] - b["cert"]).max() <= 2e-3
    assert np.array_equal(a["ts"], b["ts"])
    n_rows = a["theta"].shape[0]
    from clid_slam_amd import Mapper

    # per iteration [848 | 16 floats per row | the decoder-gradient copies of the dense exchange (clid_train_args.dec_copies)]
    dense = 3 * (848 + 16 * n_rows + Mapper.DEC_COPIES * 848)
    if sparse == "1":  # the compact exchange ran, and moved less than the dense buffer would have
        assert int(b["exchange"][1]) == 1 and 0 < int(b["exchange"][0]) < dense
        assert int(b["exchange"][2]) == int(p2p)  # over the peer-mapped buffers / over torch.distributed
        assert int(b["exchange"][3]) == int(every)
        if every == "1":  # [848 | 9 x every row] per iteration, nothing else
            assert int(b["exchange"][0]) == 3 * (848 + 9 * (n_rows - 1))
    elif mode == "numerical":
        assert int(b["exchange"][1]) == 0 and int(b["exchange"][0]) == dense


def test_p2p_timeout_falls_back_to_the_collective_on_every_rank(tmp_path):
    """ADVICE r3 (csrc/p2p.hip): a flag wait that gives up on ONE rank must not leave that rank raising and its peers hanging
    in the next collective with garbage sums applied.  The error word is agreed across the ranks at the end of the call;
    every rank restores the state it saved before the call, rules the transport out and repeats the call over the
    collective path -- the result equals the single-process one."""
    port = 29900 + (os.getpid() % 1000)
    _run(0, 1, port, str(tmp_path), "numerical")
    a = dict(np.load(os.path.join(tmp_path, "w1.npz")))
    os.environ.pop("CLID_SPARSE", None)
    mp.spawn(_run, args=(2, port, str(tmp_path), "numerical", "1", "1", "1", 1), nprocs=2, join=True)
    b = np.load(os.path.join(tmp_path, "w2.npz"))
    assert np.abs(a["loss"] - b["loss"]).max() <= 2e-6
    assert np.abs(a["theta"] - b["theta"]).max() <= 2e-5 and np.abs(a["W1"] - b["W1"]).max() <= 2e-5
    assert np.abs(a["cert"] - b["cert"]).max() <= 2e-3 and np.array_equal(a["ts"], b["ts"])
    assert int(b["exchange"][2]) == 0  # the repeated call did not use the peer-mapped buffers


def _run_size_check(rank, world, port):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import golden_io as gio
    import shim_io

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["CLID_P2P"] = "0"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    p = gio.load("pool.npz")
    g = gio.load("g6_loop_numerical_train_ln0.npz")
    cfg = shim_io.config(bs=2048)
    nm = shim_io.neural_points(cfg, base=p)
    dec = shim_io.decoder(cfg, g, "init_")
    mpr, _ = shim_io.mapper(cfg, nm, dec)
    mpr.replica_check_every = 1000  # (only the first call runs the full content check)
    idx = torch.randint(0, p["coord"].shape[0], (2, 2048), generator=torch.Generator().manual_seed(3)).cuda()
    mpr.mapping(2, index_seq=idx)
    mpr.mapping(2, index_seq=idx)  # verifies call 1's sizes (equal)
    if rank == 1:
        mpr._draw_calls = 12345    # a replica that has drawn a different number of batches
    mpr.mapping(2, index_seq=idx)  # queues the differing signature
    try:
        mpr.mapping(2, index_seq=idx)
        raised = False
    except RuntimeError as e:
        raised = "replicas diverged" in str(e)
    assert raised, rank  # EVERY rank notices (MIN / MAX over the group), one call late and without a stall
    dist.barrier()
    dist.destroy_process_group()


def test_sizes_are_cross_checked_on_every_call():
    """between two full replica checks nothing compared the ranks' sizes any more; every sharded call now carries a
    tiny asynchronous MIN / MAX of (local map size, pool size, new samples, iterations, draw counter)."""
    mp.spawn(_run_size_check, args=(2, 29950 + (os.getpid() % 40)), nprocs=2, join=True)
