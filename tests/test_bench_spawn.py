"""`python bench.py --gpus N` starts its N ranks itself (a child `torch.distributed.run`, never an exec): checked here on the
CPU with the rendezvous-only mode (`--dry-run`: gloo, no GPU work), which is all a box without a GPU can run."""
from __future__ import annotations

import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def test_gpus_2_spawns_two_ranks_and_relays_rank0_line():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--dry-run"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout  # ONE JSON line on stdout, whatever the ranks printed
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2
    assert d["cells_per_gpu"] == 129600  # the same per-GPU problem at every N (weak scaling)


def test_one_rank_has_the_same_per_gpu_size():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29611")
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--dry-run"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.strip()][-1])
    assert d["n_gpus"] == 1 and d["cells_per_gpu"] == 129600


def test_world_size_mismatch_is_refused():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "4", "--dry-run"], env=env, capture_output=True,
                       text=True, timeout=120)
    assert p.returncode != 0 and "WORLD_SIZE=2" in p.stderr


def test_parent_does_not_load_the_hip_library():
    """the spawning process must not have touched HIP: nothing of xsdba_amd is imported before the ranks are started"""
    src = (ROOT / "bench.py").read_text()
    main = src[src.index("def main():"):]
    assert main.index("spawn_ranks(args.gpus)") < main.index("from xsdba_amd import")
    head = src[:src.index("def main():")]
    assert "xsdba_amd" not in "".join(ln for ln in head.splitlines() if ln.startswith(("import", "from")))
