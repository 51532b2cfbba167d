"""Multi-GPU layout of the path: grid cells shard embarrassingly — one process per GPU, rank r owns
a contiguous block of the flattened cell index, nothing is exchanged during train/adjust, and the only
collective is the final gather of the per-rank results (RCCL over xGMI with backend "nccl"; "gloo" on CPU in
the tests).  torch.distributed is plumbing here: rendezvous, barrier, gather."""
from __future__ import annotations

import numpy as np


def shard_bounds(n_cell: int, world: int, rank: int) -> tuple[int, int]:
    """[start, stop) of rank's block; blocks differ by at most one cell and tile [0, n_cell) exactly."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of size {world}")
    base, rem = divmod(n_cell, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def shard_cells(x: np.ndarray, world: int, rank: int, cell_axis: int = 0) -> np.ndarray:
    """rank's block of a [cell, ...] array (a view)."""
    a, b = shard_bounds(x.shape[cell_axis], world, rank)
    idx = [slice(None)] * x.ndim
    idx[cell_axis] = slice(a, b)
    return x[tuple(idx)]


def gather_cells(local, n_cell: int, dst: int = 0, group=None):
    """Gather per-rank [cells_r, ...] blocks (torch tensors on the backend's device) to `dst`; returns the
    full [n_cell, ...] tensor on dst and None elsewhere.  Blocks may differ by one row, so they are padded to
    the largest block for the collective and trimmed afterwards."""
    import torch
    import torch.distributed as dist

    world, rank = dist.get_world_size(group), dist.get_rank(group)
    sizes = [shard_bounds(n_cell, world, r) for r in range(world)]
    maxrows = max(b - a for a, b in sizes)
    pad = torch.zeros((maxrows,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    outs = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    dist.gather(pad, outs, dst=dst, group=group)
    if rank != dst:
        return None
    return torch.cat([o[: b - a] for o, (a, b) in zip(outs, sizes)], dim=0)
