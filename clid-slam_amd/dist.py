"""Sharding plan of one mapping iteration over `world` GPUs.

The global batch (config.bs samples, drawn identically on every rank) is cut into contiguous
slices; the eikonal term uses every `decimation`-th sample of the GLOBAL batch
(utils/mapper.py:701-702), so a rank's first decimated sample depends on its offset; both loss
means are normalised by the GLOBAL counts so that the SUM of the ranks' gradients equals the
single-GPU gradient."""
from __future__ import annotations

from dataclasses import dataclass


@dataclass(frozen=True)
class ShardPlan:
    bs_local: int
    batch_offset: int
    fd_first: int      # local position of this rank's first decimated sample
    n_fd: int          # decimated samples owned by this rank
    n_main_global: int
    n_fd_global: int


def shard_plan(bs_global: int, world: int, rank: int, decimation: int) -> ShardPlan:
    if bs_global % world != 0:
        raise ValueError(f"batch size {bs_global} must be divisible by the world size {world}")
    bs_local = bs_global // world
    off = rank * bs_local
    r = off % decimation
    first = 0 if r == 0 else decimation - r
    n_fd = 0 if first >= bs_local else (bs_local - first + decimation - 1) // decimation
    return ShardPlan(bs_local, off, first, n_fd, bs_global, (bs_global + decimation - 1) // decimation)


def respawn_under_torchrun(script: str, argv, n: int, backend: str = "nccl") -> int:
    """`python <script> --gpus N` with no launcher around it (WORLD_SIZE unset): start the N ranks ourselves, one process
    per GPU, through `python -m torch.distributed.run` on 127.0.0.1 with a free port -- the command shape the driver
    uses for N > 1 -- and return its exit code.  With the RCCL backend fewer than N visible devices is an error (RCCL
    refuses two ranks on one device; a silent 1-rank run would report the wrong `n_gpus`); `--backend gloo` may share
    devices (dry runs of the sharded path on a 1-GPU box)."""
    import os
    import socket
    import subprocess
    import sys

    import torch

    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < 1:
        raise SystemExit(f"{os.path.basename(script)}: --gpus {n} but no GPU is visible (the HIP path has no CPU fallback)")
    if backend == "nccl" and have < n:
        raise SystemExit(f"{os.path.basename(script)}: --gpus {n} but only {have} GPU(s) visible; RCCL needs one device per "
                         "rank (use --backend gloo for a dry run of the sharded path on fewer devices)")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on these hosts (RCCL / tensor sharing)
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), script, *argv]
    return subprocess.call(cmd, env=env)
