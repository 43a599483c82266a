"""GPU: the peer-mapped gradient exchange (csrc/p2p.hip, clid_p2p_*) with several ranks on ONE device -- every rank is
its own process, the peers' buffers are mapped through HIP IPC handles exactly as on a multi-GPU node (there the loads
cross xGMI, here they stay on the device).  Sums must be the rank-order fp32 sums, bit for bit and identical on every
rank; the handles travel over gloo."""
import ctypes as C
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from clid_slam_amd import _lib

    lib = _lib.load()
    obj = _lib.p2p_exchange(dist, 3 << 20)
    result = {"made": obj is not None}
    if obj is not None:
        assert int(lib.clid_p2p_world(obj)) == world and int(lib.clid_p2p_capacity(obj)) >= (3 << 20)
        sizes = [1, 5, 848 + 9 * 1234, 100_003, (3 << 20) // 4, 64, 848 + 9 * 23_497]
        bad = 0
        for k, n in enumerate(sizes):
            g = torch.Generator().manual_seed(1000 * k)  # the SAME stream on every rank: everybody knows everybody's data
            data = [torch.randn(n, generator=g) * (10.0 ** (r - 1)) for r in range(world)]
            want = data[0].clone()
            for r in range(1, world):
                want += data[r]  # fp32, rank order
            ptr = int(lib.clid_p2p_buffer(obj))
            host = data[rank].cuda()
            _lib.check(lib.clid_debug_copy(ptr, host.data_ptr(), n * 4, _lib.stream()), "copy in")
            _lib.check(lib.clid_p2p_allreduce(obj, n, _lib.stream()), "clid_p2p_allreduce")
            out = torch.empty(n, device="cuda")
            _lib.check(lib.clid_debug_copy(out.data_ptr(), ptr, n * 4, _lib.stream()), "copy out")
            torch.cuda.synchronize()
            bad += int((out.cpu() != want).sum())
            assert int(lib.clid_p2p_buffer(obj)) != ptr  # the buffers alternate
        # bitwise OR of flag bytes held in ordinary device memory (the touched-row flags of a chunk): odd sizes, both buffers
        for k, nbytes in enumerate((1, 17, 23_497 * 3, 1_000_003)):
            g = torch.Generator().manual_seed(500 + k)
            flags = [(torch.rand(nbytes, generator=g) < 0.2).to(torch.uint8) for _ in range(world)]
            want = flags[0].clone()
            for r in range(1, world):
                want |= flags[r]
            mine = flags[rank].cuda()
            _lib.check(lib.clid_p2p_allreduce_or(obj, mine.data_ptr(), nbytes, _lib.stream()), "clid_p2p_allreduce_or")
            torch.cuda.synchronize()
            bad += int((mine.cpu() != want).sum())
        # many exchanges back to back without any host synchronisation in between (the training loop's pattern: fill the
        # current buffer, exchange, read the sums, next buffer), sizes changing every time; verified on the device
        gen = torch.Generator().manual_seed(77)
        sizes2 = torch.randint(1, 200_000, (150,), generator=gen).tolist()
        wrong = torch.zeros(1, device="cuda", dtype=torch.int64)
        base = torch.arange(200_000, device="cuda", dtype=torch.float32)
        for k, n in enumerate(sizes2):
            ptr = int(lib.clid_p2p_buffer(obj))
            mine = ((base[:n] * (k % 7 + 1) + rank * 3 + k) % 1021).contiguous()  # small integers: sums exact in any order
            _lib.check(lib.clid_debug_copy(ptr, mine.data_ptr(), n * 4, _lib.stream()), "copy in")
            _lib.check(lib.clid_p2p_allreduce(obj, n, _lib.stream()), "clid_p2p_allreduce")
            out = torch.empty(n, device="cuda")
            _lib.check(lib.clid_debug_copy(out.data_ptr(), ptr, n * 4, _lib.stream()), "copy out")
            want = sum(((base[:n] * (k % 7 + 1) + r * 3 + k) % 1021) for r in range(world))
            wrong += (out != want).sum()
        torch.cuda.synchronize()
        bad += int(wrong.item())
        _lib.check(lib.clid_p2p_status(obj, _lib.stream()), "clid_p2p_status")
        result["bad"] = bad
    np.save(os.path.join(out_dir, f"r{rank}.npy"), np.array([int(result["made"]), result.get("bad", -1)]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_peer_mapped_allreduce_is_the_rank_order_sum_on_every_rank(tmp_path, world):
    port = 29900 + (os.getpid() % 500) + world
    mp.spawn(_run, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    got = [np.load(os.path.join(tmp_path, f"r{r}.npy")) for r in range(world)]
    assert all(int(g[0]) == 1 for g in got), "the exchange object could not be set up (IPC mapping or self-test failed)"
    assert all(int(g[1]) == 0 for g in got)


def test_single_rank_object_is_a_no_op():
    sys.path.insert(0, ROOT)
    from clid_slam_amd import _lib

    lib = _lib.load()
    nb = int(lib.clid_p2p_blob_bytes())
    blob = (C.c_uint8 * nb)()
    obj = C.c_void_p()
    _lib.check(lib.clid_p2p_create(0, 1, 1 << 20, C.byref(obj), blob), "clid_p2p_create")
    try:
        _lib.check(lib.clid_p2p_selftest(obj, _lib.stream()), "clid_p2p_selftest")
        a = int(lib.clid_p2p_buffer(obj))
        _lib.check(lib.clid_p2p_allreduce(obj, 1000, _lib.stream()), "clid_p2p_allreduce")
        assert int(lib.clid_p2p_buffer(obj)) != a
        assert lib.clid_p2p_allreduce(obj, (1 << 20), _lib.stream()) != 0  # 4 MiB > capacity
        _lib.check(lib.clid_p2p_status(obj, _lib.stream()), "clid_p2p_status")
    finally:
        lib.clid_p2p_destroy(obj)
