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
