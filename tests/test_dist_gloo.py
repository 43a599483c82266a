"""World-size-2 (gloo, CPU) check of the data-parallel scheme of SURVEY.md section 8e: each rank takes
its contiguous slice of the GLOBAL batch (clid_slam_amd.dist.shard_plan: decimation phase and loss
normalisers), gradients are summed with all_reduce, and the result equals the single-process gradient
on the full batch.  The per-shard compute is the CPU oracle (the HIP kernels take exactly these
plan values as batch_offset / inv_n_main / inv_n_eik)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, bs, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import golden_io as gio
    from clid_slam_amd.dist import shard_plan
    from oracle import cpu_ref as O

    torch.set_num_threads(2)
    st = gio.map_state()
    dec = gio.decoder()
    pool, _ = gio.sample_pool()
    gen = torch.Generator().manual_seed(7)  # same seed on every rank => same global batch
    index = torch.randint(0, pool.global_coord.shape[0], (bs,), generator=gen)
    plan = shard_plan(bs, world, rank, 10)
    lc = O.LoopConfig(fd_first=plan.fd_first, loss_scale_counts=(plan.n_main_global, plan.n_fd_global))
    out = O.loss_and_grads(st, dec, pool, index[plan.batch_offset : plan.batch_offset + plan.bs_local], lc)
    flat = torch.cat([out["grad_W1"].flatten(), out["grad_b1"], out["grad_W2"].flatten(), out["grad_b2"],
                      out["grad_theta"].flatten(), out["loss"].reshape(1)])
    dist.all_reduce(flat)
    cert = st.local_point_certainties - gio.map_state().local_point_certainties
    dist.all_reduce(cert)
    ts = st.local_point_ts_update.clone()
    dist.all_reduce(ts, op=dist.ReduceOp.MAX)
    if rank == 0:
        np.savez(os.path.join(out_dir, "dp.npz"), flat=flat.numpy(), cert=cert.numpy(), ts=ts.numpy(), index=index.numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("bs", [2048, 1030])
def test_two_rank_sharded_step_equals_single_process(tmp_path, bs):
    import golden_io as gio
    from oracle import cpu_ref as O

    port = 29500 + (os.getpid() % 2000) + (0 if bs == 2048 else 1)
    mp.spawn(_worker, args=(2, port, bs, str(tmp_path)), nprocs=2, join=True)
    z = np.load(os.path.join(tmp_path, "dp.npz"))
    st = gio.map_state()
    st0 = gio.map_state()
    dec = gio.decoder()
    pool, _ = gio.sample_pool()
    out = O.loss_and_grads(st, dec, pool, torch.from_numpy(z["index"]), O.LoopConfig())
    ref = torch.cat([out["grad_W1"].flatten(), out["grad_b1"], out["grad_W2"].flatten(), out["grad_b2"],
                     out["grad_theta"].flatten(), out["loss"].reshape(1)]).numpy()
    scale = np.abs(ref).max()
    assert np.abs(z["flat"] - ref).max() <= 2e-6 * max(scale, 1.0)
    assert abs(z["flat"][-1] - ref[-1]) <= 1e-6  # the summed shard losses are the global loss
    cert_ref = (st.local_point_certainties - st0.local_point_certainties).numpy()
    assert np.abs(z["cert"] - cert_ref).max() <= 1e-4
    assert np.array_equal(z["ts"], st.local_point_ts_update.numpy())


def test_shard_plan_partitions_the_decimation_lattice():
    from clid_slam_amd.dist import shard_plan

    for bs, world, decim in ((16384 * 8, 8, 10), (4096, 4, 10), (2060, 2, 10), (64, 2, 1), (90, 3, 7)):
        plans = [shard_plan(bs, world, r, decim) for r in range(world)]
        assert sum(p.bs_local for p in plans) == bs
        lattice = set(range(0, bs, decim))
        got = set()
        for p in plans:
            mine = {p.batch_offset + p.fd_first + k * decim for k in range(p.n_fd)}
            assert all(p.batch_offset <= q < p.batch_offset + p.bs_local for q in mine)
            got |= mine
        assert got == lattice and plans[0].n_fd_global == len(lattice)
    with pytest.raises(ValueError):
        shard_plan(100, 3, 0, 10)
