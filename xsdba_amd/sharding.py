"""Multi-GPU layout of the path: grid cells shard embarrassingly — one process per GPU, rank r owns a
contiguous block of the flattened cell index (the role dask chunks over lat / lon play for the reference: map_blocks,
base.py:640-650, 715), nothing is exchanged during train / adjust, and the only collective is the final gather of the
per-rank results.

The gather itself is native: ``xsdba_hip_gather`` posts RCCL sends / receives straight from / into the device buffers on
the library's stream (no staging copy, no torch tensor).  What the caller's launcher contributes is the rendezvous only —
handing rank 0's 128-byte RCCL token to the other ranks: ``init_comm`` does it through ``torch.distributed`` when a
process group is up (``bench.py`` under ``torch.distributed.run``), through a file otherwise.

``shard_bounds`` / ``shard_cells`` are the host-side view of the same plan (``xsdba_hip_shard_plan``);
``gather_cells`` is the torch.distributed fallback used on CPU (``gloo``) by the tests.
"""
from __future__ import annotations

import ctypes as C
import os
import time

import numpy as np

COMM_ID_BYTES = 128


def shard_bounds(n_cell: int, world: int, rank: int) -> tuple[int, int]:
    """[start, stop) of rank's block; blocks differ by at most one cell and tile [0, n_cell) exactly
    (the arithmetic of ``xsdba_hip_shard_plan``)."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of size {world}")
    base, rem = divmod(n_cell, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def shard_plan(n_cell: int, world: int, rank: int) -> tuple[int, int]:
    """(first cell, number of cells) of `rank` from the library itself (no GPU needed)"""
    from ._lib import check, load

    first, count = C.c_int64(0), C.c_int64(0)
    check(load().xsdba_hip_shard_plan(C.c_int64(n_cell), C.c_int(world), C.c_int(rank), C.byref(first), C.byref(count)), "shard_plan")
    return first.value, count.value


def shard_cells(x: np.ndarray, world: int, rank: int, cell_axis: int = 0) -> np.ndarray:
    """rank's block of a [cell, ...] array (a view)."""
    a, b = shard_bounds(x.shape[cell_axis], world, rank)
    idx = [slice(None)] * x.ndim
    idx[cell_axis] = slice(a, b)
    return x[tuple(idx)]


def _job_nonce(job_id) -> bytes:
    """16 bytes every rank of ONE job derives alike and NO other job does: from `job_id`, else from XSDBA_AMD_JOB_ID, else from
    the launcher's run id (TORCHELASTIC_RUN_ID, unless it is torchrun's constant default "none").  Nothing else will do: a
    rendezvous address (MASTER_ADDR:MASTER_PORT) is the same for consecutive jobs on the default port, and a rank that starts
    polling before rank 0 has removed a crashed earlier job's file would take that job's dead token and hang in comm_init —
    so without a per-job id the file path is refused (ValueError) rather than guessed."""
    import hashlib

    if job_id is None:
        env = os.environ
        job_id = env.get("XSDBA_AMD_JOB_ID")
        if not job_id:
            rid = env.get("TORCHELASTIC_RUN_ID")
            job_id = rid if rid and rid.lower() != "none" else None
    if job_id is None or str(job_id) == "":
        raise ValueError("the token-file rendezvous needs a per-job id shared by the ranks: pass job_id=..., or set XSDBA_AMD_JOB_ID "
                         "(a launcher's unique run id works too), or initialise torch.distributed and let the token travel there")
    return hashlib.sha256(str(job_id).encode()).digest()[:16]


def _token_through_file(token: bytes, rank: int, token_file: str, nonce: bytes, timeout: float) -> bytes:
    """Rank 0 publishes `token`, the others return rank 0's.  The file holds the 16-byte job nonce + the token.  Freshness is
    never judged by clocks (ranks start seconds apart, file-system and local clocks differ): rank 0 removes whatever is there
    before it writes (atomically) and again once every rank has joined; the others take the first complete file that carries
    this job's nonce — a leftover of another job is ignored, however recent."""
    if rank == 0:
        try:
            os.unlink(token_file)
        except FileNotFoundError:
            pass
        tmp = f"{token_file}.{os.getpid()}.tmp"
        with open(tmp, "wb") as f:
            f.write(nonce + token)
        os.replace(tmp, token_file)
        return token
    t0 = time.monotonic()
    while True:
        try:
            with open(token_file, "rb") as f:
                raw = f.read(16 + COMM_ID_BYTES + 1)
            if len(raw) == 16 + COMM_ID_BYTES and raw[:16] == nonce:
                return raw[16:]
        except FileNotFoundError:
            pass
        if time.monotonic() - t0 > timeout:
            raise TimeoutError(f"no RCCL token of this job at {token_file} after {timeout:.0f} s")
        time.sleep(0.01)


def init_comm(ctx, rank: int, world: int, token_file: str | None = None, job_id=None, timeout: float = 600.0) -> None:
    """Collective: set up the RCCL communicator of `ctx`.  Rank 0 creates the token; it travels through
    torch.distributed when a process group is initialised, else through `token_file` (written atomically by rank 0, polled by
    the others — any shared path works: the ranks of one node share /tmp).  `job_id` (or XSDBA_AMD_JOB_ID / the launcher's
    run id in the environment) tells this job's token from one a crashed earlier job left at the same path."""
    from ._lib import check, load

    buf = (C.c_char * COMM_ID_BYTES)()
    dist = None
    try:
        import torch.distributed as dist  # noqa: F811

        if not (dist.is_available() and dist.is_initialized()):
            dist = None
    except Exception:  # noqa: BLE001
        dist = None
    if rank == 0:
        check(load().xsdba_hip_comm_unique_id(buf), "comm_unique_id")
    if world > 1:
        if dist is not None:
            box = [bytes(buf.raw) if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            buf.raw = box[0]
        else:
            if not token_file:
                raise ValueError("init_comm needs torch.distributed to be initialised or a token_file shared by the ranks")
            buf.raw = _token_through_file(bytes(buf.raw), rank, token_file, _job_nonce(job_id), timeout)
    check(load().xsdba_hip_comm_init(ctx.handle, buf, C.c_int(rank), C.c_int(world)), "comm_init")
    if world > 1 and dist is None and rank == 0:
        try:
            os.unlink(token_file)  # every rank has joined: the token has served
        except OSError:
            pass


def comm_info(ctx) -> tuple[int, int]:
    """(rank, number of ranks) as RCCL reports them for the communicator of `ctx`"""
    from ._lib import check, load

    r, n = C.c_int(-1), C.c_int(0)
    check(load().xsdba_hip_comm_info(ctx.handle, C.byref(r), C.byref(n)), "comm_info")
    return r.value, n.value


def gather_device(ctx, local, n_cell: int, root: int = 0, out=None):
    """Native gather of the per-rank ``[cells_r, row...]`` DeviceArrays to `root`: returns the full ``[n_cell, row...]``
    DeviceArray on the root (``out`` when given) and None elsewhere.  Asynchronous on the context's stream."""
    from ._lib import check, dtype_code, load

    rank, world = comm_info(ctx)
    row = int(np.prod(local.shape[1:], dtype=np.int64)) if len(local.shape) > 1 else 1
    full = None
    if rank == root:
        full = out if out is not None else ctx.empty((n_cell,) + tuple(local.shape[1:]), local.dtype)
    check(load().xsdba_hip_gather(ctx.handle, C.c_void_p(local.ptr), C.c_int64(local.shape[0]), C.c_int64(row),
                                  C.c_int(dtype_code(local.dtype)), C.c_void_p(full.ptr) if full is not None else None,
                                  C.c_int64(n_cell), C.c_int(root)), "gather")
    return full


def gather_cells(local, n_cell: int, dst: int = 0, group=None):
    """torch.distributed fallback (CPU / gloo in the tests): gather per-rank [cells_r, ...] tensors to `dst`; returns the
    full [n_cell, ...] tensor on dst and None elsewhere.  Blocks may differ by one row, so they are padded to the largest
    block for the collective and trimmed afterwards."""
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
