"""N>1 path on CPU: world-size-2 gloo processes shard the cells, work on their own block and gather."""
from __future__ import annotations

import os
import subprocess
import sys
import textwrap
from pathlib import Path

import numpy as np
import pytest

from xsdba_amd.sharding import shard_bounds, shard_cells, shard_plan

ROOT = Path(__file__).resolve().parent.parent


@pytest.mark.parametrize("n,world", [(10, 1), (10, 2), (10, 3), (7, 8), (64800, 8), (1036800, 8)])
def test_shard_bounds_tile_the_cells(n, world):
    blocks = [shard_bounds(n, world, r) for r in range(world)]
    assert blocks[0][0] == 0 and blocks[-1][1] == n
    assert all(blocks[i][1] == blocks[i + 1][0] for i in range(world - 1))
    sizes = [b - a for a, b in blocks]
    assert max(sizes) - min(sizes) <= 1
    x = np.arange(n * 3).reshape(n, 3)
    np.testing.assert_array_equal(np.concatenate([shard_cells(x, world, r) for r in range(world)]), x)
    with pytest.raises(ValueError):
        shard_bounds(n, world, world)
    # the library's own plan (xsdba_hip_shard_plan: pure arithmetic, callable without a GPU) is the same tiling
    assert [shard_plan(n, world, r) for r in range(world)] == [(a, b - a) for a, b in blocks]


def test_two_rank_gloo_shard_and_gather(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(textwrap.dedent(f"""
        import os, sys
        sys.path.insert(0, {str(ROOT)!r})
        import numpy as np, torch, torch.distributed as dist
        from xsdba_amd.sharding import shard_bounds, shard_cells, shard_plan, gather_cells
        dist.init_process_group("gloo")
        rank, world = dist.get_rank(), dist.get_world_size()
        n_cell, T = 11, 5
        full = np.arange(n_cell * T, dtype=np.float32).reshape(n_cell, T)
        mine = shard_cells(full, world, rank)
        # stand-in for the per-cell hot path (cells are independent): any pure function of a cell's own series
        local = torch.from_numpy(mine * 2 + 1)
        dist.barrier()
        out = gather_cells(local, n_cell, dst=0)
        if rank == 0:
            np.testing.assert_array_equal(out.numpy(), full * 2 + 1)
            print("GATHER_OK", tuple(out.shape))
        else:
            assert out is None
        # max-over-ranks timing reduction used by bench.py
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert t.item() == world
        dist.destroy_process_group()
    """))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    proc = subprocess.run(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
         "--master-port", "29653", str(script)],
        capture_output=True, text=True, env=env, timeout=300)
    assert proc.returncode == 0, proc.stdout + proc.stderr
    assert "GATHER_OK (11, 5)" in proc.stdout


def test_token_file_rendezvous_ignores_clocks_and_other_jobs(tmp_path):
    """xsdba_amd.sharding._token_through_file: a rank that arrives long after rank 0 wrote still gets the token (no mtime
    test), and a complete file left by ANOTHER job is never taken for this one's."""
    import os
    import threading
    import time

    from xsdba_amd import sharding as sh

    f = str(tmp_path / "tok")
    tok = bytes(range(sh.COMM_ID_BYTES % 256)) * 1 + bytes(sh.COMM_ID_BYTES - (sh.COMM_ID_BYTES % 256))
    tok = tok[:sh.COMM_ID_BYTES]
    n_a, n_b = sh._job_nonce("job-a"), sh._job_nonce("job-b")
    assert n_a != n_b and len(n_a) == 16
    # rank 0 of job a wrote long ago (mtime far in the past): a late rank still accepts it
    assert sh._token_through_file(tok, 0, f, n_a, 5.0) == tok
    os.utime(f, (time.time() - 3600, time.time() - 3600))
    assert sh._token_through_file(b"", 1, f, n_a, 5.0) == tok
    # a rank of job b waits past job a's leftover until its own rank 0 has written
    other = bytes(reversed(tok))
    got = {}
    th = threading.Thread(target=lambda: got.setdefault("t", sh._token_through_file(b"", 1, f, n_b, 10.0)))
    th.start()
    time.sleep(0.3)
    assert th.is_alive()
    sh._token_through_file(other, 0, f, n_b, 5.0)
    th.join(10)
    assert got["t"] == other
    with pytest.raises(TimeoutError):
        sh._token_through_file(b"", 1, f, sh._job_nonce("job-c"), 0.2)


def test_token_file_rendezvous_needs_a_per_job_id(monkeypatch):
    """a rendezvous address is not a per-job secret (consecutive jobs share the default port), torchrun's default run id is the
    constant "none": neither may key the token file — a stale token of a crashed job would be accepted"""
    from xsdba_amd import sharding as sh

    for var in ("XSDBA_AMD_JOB_ID", "TORCHELASTIC_RUN_ID"):
        monkeypatch.delenv(var, raising=False)
    monkeypatch.setenv("MASTER_ADDR", "127.0.0.1")
    monkeypatch.setenv("MASTER_PORT", "29500")
    with pytest.raises(ValueError, match="per-job id"):
        sh._job_nonce(None)
    monkeypatch.setenv("TORCHELASTIC_RUN_ID", "none")
    with pytest.raises(ValueError):
        sh._job_nonce(None)
    monkeypatch.setenv("TORCHELASTIC_RUN_ID", "6f1c0c2e-77aa")
    a = sh._job_nonce(None)
    monkeypatch.setenv("XSDBA_AMD_JOB_ID", "slurm-4711")
    b = sh._job_nonce(None)
    assert a != b and b == sh._job_nonce("slurm-4711") and len(a) == 16
