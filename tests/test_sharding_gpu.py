"""Multi-GPU path on real hardware, as far as one GPU goes:
  * the native gather (xsdba_hip_gather: RCCL from / into device buffers) with a one-rank communicator;
  * two processes that each run the REAL hot path (EQM train + adjust through the C ABI) on their shard of the cells and
    gather the result — over gloo here, because RCCL refuses two ranks on one device; the 8-GPU run is the driver's."""
from __future__ import annotations

import os
import subprocess
import sys
import textwrap
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def test_native_gather_single_rank():
    from xsdba_amd import hip
    from xsdba_amd.sharding import comm_info, gather_device, init_comm

    ctx = hip.Context(0)
    try:
        assert comm_info(ctx) == (-1, 0)
        init_comm(ctx, 0, 1)
        assert comm_info(ctx) == (0, 1)
        x = np.arange(37 * 100, dtype=np.float32).reshape(37, 100)
        d = ctx.to_device(x)
        full = gather_device(ctx, d, 37, root=0)
        ctx.synchronize()
        np.testing.assert_array_equal(full.download(), x)
        with pytest.raises(Exception):
            gather_device(ctx, d, 38, root=0)  # the block does not match the plan
    finally:
        ctx.close()


def test_two_processes_real_hot_path_and_gather(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(textwrap.dedent(f"""
        import os, sys
        sys.path.insert(0, {str(ROOT)!r})
        import numpy as np, torch, torch.distributed as dist
        from xsdba_amd import hip
        from xsdba_amd.sharding import shard_cells, gather_cells, shard_plan
        dist.init_process_group("gloo")
        rank, world = dist.get_rank(), dist.get_world_size()
        rng = np.random.default_rng(0)
        n_cell, T = 21, 3650
        gidx = (np.arange(T) % 365).astype(np.int32)
        q = ((np.arange(20) + 0.5) / 20).astype(np.float32)
        ref = (280 + 10 * rng.standard_normal((n_cell, T))).astype(np.float32)
        hist = (281 + 9 * rng.standard_normal((n_cell, T))).astype(np.float32)
        sim = (hist + 1).astype(np.float32)
        a, n = shard_plan(n_cell, world, rank)
        assert shard_cells(ref, world, rank).shape[0] == n
        af, hq = hip.eqm_train(ref[a:a + n], hist[a:a + n], gidx, 365, 31, q, "+")
        scen = hip.qm_adjust(sim[a:a + n], af, hq, gidx, "nearest", "constant", "+")
        out = gather_cells(torch.from_numpy(scen), n_cell, dst=0)
        if rank == 0:
            af0, hq0 = hip.eqm_train(ref, hist, gidx, 365, 31, q, "+")
            exp = hip.qm_adjust(sim, af0, hq0, gidx, "nearest", "constant", "+")
            np.testing.assert_array_equal(out.numpy(), exp)
            print("SHARDED_HOT_PATH_OK", tuple(out.shape))
        dist.barrier()
        dist.destroy_process_group()
    """))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    proc = subprocess.run(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
         "--master-port", "29671", str(script)],
        capture_output=True, text=True, env=env, timeout=600)
    assert proc.returncode == 0, proc.stdout + proc.stderr
    assert "SHARDED_HOT_PATH_OK (21, 3650)" in proc.stdout
