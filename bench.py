#!/usr/bin/env python3
"""bench.py — grid-cells/s of the quantile-mapping hot path on MI355X (contract: see the task brief).

One "step" = one full train + adjust pass of the workload over the rank's block of grid cells, with the
synthetic (cell, time) cubes already resident in HBM (generated on the device).  Default
workload = the one BASELINE.json's metric names: EmpiricalQuantileMapping, Grouper("time.dayofyear",
window=31), kind "+", 30 years of daily data (10950 noleap days), nquantiles 20, interp "nearest", on the
per-GPU shard of north_star's 1440x720 cube over 8 GPUs (129 600 cells per GPU at EVERY N, so that the N = 1, 2, 4, 8
series is weak scaling of one per-GPU problem; every rank owns its own block of cells, no data-path collective; --cells
64800 / 259200 give the 360x180 and 720x360 grids of configs[1] / [2]).  Other workloads: --workload eqm_month | dqm_doy31 | qdm_doy31 | eqm_time | dqm_doy31_nan
(BASELINE configs[4]: use --days 21900).

N > 1: launched by torch.distributed.run, one process per GPU; torch is used only for the rendezvous, the barrier and the
MAX over ranks (CPU tensors over gloo); the one collective of the path, the final gather of scen (--gather, timed
separately and verified), is the library's own RCCL call over xGMI, straight from the device buffers.
"""
from __future__ import annotations

import argparse
import json
import os
import re
import sys
import threading
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 TB/s achievable copy)

WORKLOADS = {
    # name: (method, group, window, kind, generator)
    "eqm_doy31": ("eqm", "dayofyear", 31, "+", 0),
    "eqm_month": ("eqm", "month", 1, "+", 0),
    "eqm_time": ("eqm", "time", 1, "+", 0),
    "dqm_doy31": ("dqm", "dayofyear", 31, "+", 0),
    "qdm_doy31": ("qdm", "dayofyear", 31, "*", 1),
    "qdm_doy31_rw": ("qdm", "dayofyear", 31, "*", 1),  # BASELINE configs[3], second mode: rank_window=True
    "qdm_month": ("qdm", "month", 1, "*", 1),  # QDM over un-windowed groups (ranks straight from the sorted order)
    # BASELINE configs[4]: 60 years daily, 30 % of the cells all-NaN ("ocean") + 1 % i.i.d. NaN elsewhere (run it with
    # --days 21900; cells per GPU as memory allows)
    "dqm_doy31_nan": ("dqm", "dayofyear", 31, "+", 0),
}
GRID_NAMES = {129600: " (1/8 of the 1440x720 grid)", 64800: " (the 360x180 grid)", 259200: " (the 720x360 grid)"}
DEFAULT_CELLS_PER_GPU = 1440 * 720 // 8  # 129 600: the per-GPU shard of north_star's 1440x720 cube over 8 GPUs
NAN_MASK = {"dqm_doy31_nan": (0.30, 0.01)}  # workload -> (fraction of all-NaN cells, i.i.d. NaN fraction elsewhere)


def group_index(group: str, T: int):
    d0 = np.arange(T) % 365
    if group == "dayofyear":
        return d0.astype(np.int32), 365
    if group == "month":
        cum = np.concatenate([[0], np.cumsum([31, 28, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31])])
        return (np.searchsorted(cum, d0, side="right") - 1).astype(np.int32), 12
    return np.zeros(T, np.int32), 1


def algorithmic_bytes_per_cell(method, T, G, Q, itemsize=4, trend_array=True):
    """compulsory HBM traffic per grid cell (train, adjust).  trend_array=False: the DQM adjustment that does not
    write the [cell, time] trend array (what the timed step runs: the mapping kernel evaluates the trend line itself)."""
    train = 2 * T * itemsize + 2 * G * Q * itemsize + (G * itemsize if method == "dqm" else 0)
    if method == "qdm":
        adjust = T * itemsize + G * Q * itemsize + T * itemsize
    else:
        adjust = T * itemsize + 2 * G * Q * itemsize + T * itemsize
        if method == "dqm":
            adjust += G * itemsize + (T * itemsize if trend_array else 0)
    return train, adjust


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="eqm_doy31", choices=sorted(WORKLOADS))
    ap.add_argument("--cells", type=int, default=None, help="grid cells per GPU (default at every N: 129 600, an eighth of "
                    "north_star's 1440x720 grid; 64 800 = the 360x180 grid, 259 200 = the 720x360 one)")
    ap.add_argument("--days", type=int, default=10950)
    ap.add_argument("--nquantiles", type=int, default=20)
    ap.add_argument("--time-major", action="store_true", help="store the cubes as (time, cell) instead of (cell, time)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-api-level", action="store_true", help="skip the secondary figure through the drop-in API")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target duration of the CPU baseline sample")
    ap.add_argument("--gather", action="store_true", help="also time (and verify) the final native RCCL gather of scen to rank 0 "
                    "(the default with more than one rank)")
    ap.add_argument("--no-gather", action="store_true", help="N > 1: skip the RCCL gather of scen")
    ap.add_argument("--gather-timeout", type=float, default=90.0, help="seconds the gather leg may take before the line is printed without it")
    ap.add_argument("--dry-run", action="store_true", help=argparse.SUPPRESS)  # rendezvous only, no GPU work (tests/test_bench_spawn.py)
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N`: start the N ranks ourselves.  Nothing has touched HIP in this process (the library is
        # imported below, in the ranks), and the ranks are CHILD processes — never an exec of this one.
        raise SystemExit(spawn_ranks(args.gpus))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.cells is None:
        args.cells = DEFAULT_CELLS_PER_GPU  # the same per-GPU problem at every N: the series over N is weak scaling
    if world > 1 and not args.no_gather:
        args.gather = True
    if args.dry_run:
        return dry_run(world, rank, args.cells)

    dist = None
    # torch.distributed is rendezvous / barrier / MAX-over-ranks plumbing only, on CPU tensors ("gloo"): the process then
    # runs ONE HIP runtime — the system one this library is built on — and the data-path collective (the final gather) is the
    # library's own RCCL call (xsdba_hip_gather).  XSDBA_BENCH_BACKEND=nccl routes the plumbing through torch's bundled RCCL
    # instead (a second HIP runtime in the process).
    backend = os.environ.get("XSDBA_BENCH_BACKEND", "gloo")
    from xsdba_amd import _lib as xlib

    ndev = xlib.device_count()
    if world > max(ndev, 1) and not os.environ.get("XSDBA_BENCH_SHARE_GPU"):
        raise SystemExit(f"{world} ranks but {ndev} visible GPU(s): one process per GPU (XSDBA_BENCH_SHARE_GPU=1 to rehearse on fewer)")
    device = local_rank % max(ndev, 1)
    if world > 1:
        import torch
        import torch.distributed as dist

        if backend == "nccl":
            torch.cuda.set_device(device)
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group(backend)
    tdev = "cuda" if backend == "nccl" else "cpu"

    from xsdba_amd import hip

    ctx = hip.Context(device)
    method, group, W, kind, gen = WORKLOADS[args.workload]
    rank_window = args.workload.endswith("_rw")
    C, T, Q = args.cells, args.days, args.nquantiles
    gidx, G = group_index(group, T)
    q = ((np.arange(Q) + 0.5) / Q).astype(np.float32)
    tnum = (np.arange(T, dtype=np.float64) + 10950.0) * 86400e9
    seed = {"eqm_time": 0, "eqm_month": 1, "dqm_doy31": 2, "qdm_doy31": 3, "qdm_doy31_rw": 3, "dqm_doy31_nan": 4, "qdm_month": 5}.get(args.workload, 1)
    cell0 = rank * C
    ncf, nif = NAN_MASK.get(args.workload, (0.0, 0.0))
    ref = ctx.synth((C, T), np.float32, gen, 0, seed, cell0, ncf, nif, time_major=args.time_major)
    hist = ctx.synth((C, T), np.float32, gen, 1, seed, cell0, ncf, nif, time_major=args.time_major)
    sim = ctx.synth((C, T), np.float32, gen, 2, seed, cell0, ncf, nif, time_major=args.time_major)
    ctx.synchronize()

    # output buffers are allocated once and reused by every step (a production pipeline does the same; a fresh
    # multi-GB hipMalloc per step costs milliseconds of driver page clearing inside the step)
    f32 = np.float32
    o_af, o_hq = ctx.empty((C, G, Q), f32), ctx.empty((C, G, Q), f32)
    o_scen = ctx.empty((C, T), f32)
    o_sc = ctx.empty((C, G), f32) if method == "dqm" else None
    ctx.synchronize()

    # scen in the layout of sim: the time-major leg (below), and --time-major itself on one GPU (the N > 1 gather verifies
    # cell-major blocks)
    tm_out = bool(args.time_major) and method == "eqm" and G > 1 and world == 1

    def step():
        """one pass of the hot path over this rank's cells; returns the device outputs"""
        if method == "dqm":
            af, hq, sc = hip.dqm_train(ref, hist, gidx, G, W, q, kind, ctx=ctx, out=(o_af, o_hq, o_sc))
            # (what DetrendedQuantileMapping.adjust runs: scen only — the trend is a line per group, evaluated by the mapping kernel)
            scen, _ = hip.dqm_adjust(sim, af, hq, sc, gidx, W, tnum, 1, "nearest", "constant", kind, ctx=ctx,
                                     out=(o_scen, None), want_trend=False)
            return [af, hq, sc, scen]
        af, hq = hip.eqm_train(ref, hist, gidx, G, W, q, kind, ctx=ctx, out=(o_af, o_hq))
        if method == "qdm":
            scen, _ = hip.qdm_adjust(sim, af, q, gidx, W, rank_window, "nearest", "constant", kind, want_sim_q=False, ctx=ctx,
                                     out=o_scen)
        elif tm_out:
            # (time, cell) cubes: sim is mapped in place, scen written (time, cell) like it (xsdba_amd/csrc/adjust_tm.hpp)
            scen = hip.qm_adjust(sim, af, hq, gidx, "nearest", "constant", kind, ctx=ctx, out=o_scen.reshape(T, C), out_time_major=True)
        else:
            scen = hip.qm_adjust(sim, af, hq, gidx, "nearest", "constant", kind, ctx=ctx, out=o_scen)
        return [af, hq, scen]

    def free(outs):
        pass

    def barrier():
        ctx.synchronize()
        if dist is not None:
            import torch

            if tdev == "cuda":
                torch.cuda.synchronize()
            dist.barrier()

    for _ in range(args.warmup):
        free(step())
    barrier()
    ctx.profile_enable(True)
    ctx.profile_read(reset=True)
    t0 = time.perf_counter()
    outs = None
    for _ in range(args.steps):
        if outs is not None:
            free(outs)
        outs = step()
    ctx.synchronize()
    if dist is not None and tdev == "cuda":
        import torch

        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    prof = ctx.profile_read(reset=True)
    ctx.profile_enable(False)
    if dist is not None:
        import torch

        tt = torch.tensor([elapsed], dtype=torch.float64, device=tdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        cells_total = C * world
        value = cells_total / (elapsed / args.steps)
        tb, ab = algorithmic_bytes_per_cell(method, T, G, Q, trend_array=False)  # (the timed DQM step writes no trend array)
        # dominant kernel: the one with the largest total time inside the timed region (HIP events on the
        # library's stream, recorded around every launch)
        kern = {k: v for k, v in prof.items() if v["count"] > 0}
        dom = max(kern, key=lambda k: kern[k]["total_ms"])
        dom_avg_ms = kern[dom]["total_ms"] / kern[dom]["count"]
        dom_bytes = C * (tb if dom.startswith("k_train") else ab)  # algorithmic bytes one launch moves
        achieved = dom_bytes / (dom_avg_ms * 1e-3) / 1e9
        traffic, traffic_source = pmc_traffic(args.workload, C, T, dom)
        roof = {
            "bound": "hbm", "kernel": dom, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic, "traffic_source": traffic_source,
            "avg_launch_ms": round(dom_avg_ms, 4), "algorithmic_bytes_per_launch": dom_bytes,
            "kernels": {k: {"count": v["count"], "avg_ms": round(v["total_ms"] / v["count"], 4)} for k, v in kern.items()},
        }
        if dom.startswith("k_train") and W > 1:
            # secondary figure of SURVEY §8d, NOT the roofline fraction: the bytes the reference's formulation of the same
            # step touches (every element enters `window` group rows: 2*T*W*itemsize per cell), as a rate over the same launch
            wb = C * 2 * T * W * 4
            roof["windowed_bytes_equivalent"] = {"bytes_per_launch": wb, "GB/s": round(wb / (dom_avg_ms * 1e-3) / 1e9, 1),
                                                 "note": "rows are never materialised: one sort per series + rank-space selection"}
        out = {
            "metric": "grid-cells/sec for EQM train+adjust, DOY-grouped, 30yr daily; 1/2/4/8 GPU",
            "value": round(value, 2), "unit": "grid-cells/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic (device-generated, counter-based)",
            "config": {
                "workload": f"{args.workload}: {method.upper()} train+adjust{' (rank_window=True)' if rank_window else ''}, group={group}, window={W}, kind='{kind}', "
                            f"nquantiles={Q}, interp=nearest, {C} cells/GPU x {T} days fp32, "
                            f"{'time-major (time,cell)' if args.time_major else 'cell-major (cell,time)'} cubes resident in HBM"
                            f"{GRID_NAMES.get(C, '')}",
                "cells_per_gpu": C, "days": T, "groups": G, "window": W, "parallelism": f"cells sharded over {world} GPU(s)",
            },
            "roofline": roof,
        }
        if not args.no_api_level and not args.time_major and world == 1:
            out["api_level"] = api_level(ctx, ref, hist, sim, method, group, W, kind, Q, C, T, rank_window)
            # secondary figure: the same step on (time, cell) cubes — what a (time, lat, lon) array is — through the same entry
            # points (the library re-lays every series with a tiled transpose on the way in and out, charged to the step)
            tm = [ctx.synth((C, T), np.float32, gen, i, seed, cell0, ncf, nif, time_major=True) for i in range(3)]
            cm = (ref, hist, sim)
            ref, hist, sim = tm
            tm_out = method == "eqm" and G > 1
            step()
            ctx.synchronize()
            t1 = time.perf_counter()
            for _ in range(3):
                step()
            ctx.synchronize()
            dt_tm = (time.perf_counter() - t1) / 3
            out["time_major"] = {"value": round(C / dt_tm, 2), "unit": "grid-cells/s", "ms_per_step": round(dt_tm * 1e3, 3),
                                 "vs_cell_major": round((C / dt_tm) / value, 4),
                                 "note": "inputs (time, cell); scen is written " + ("(time, cell)" if tm_out else "(cell, time)") +
                                         "; never the reported `value`"}
            tm_out = False
            ref, hist, sim = cm
            for a in tm:
                a.free()
        if not args.no_cpu_baseline and world == 1:  # (rank 0 at N = 1 only)
            out["cpu_baseline"] = cpu_baseline(ctx, hip, ref, hist, sim, method, gidx, G, W, q, kind, tnum, C, T,
                                               args.cpu_seconds, args.time_major, rank_window,
                                               synth_args=(gen, seed, cell0, ncf, nif))
    else:
        out = None

    if args.gather:
        # the one collective of the path, last: the final gather of the per-rank scen blocks to rank 0 (timed separately, never
        # part of `value`).  Guarded: a rank that fails or a collective that does not come back within --gather-timeout must
        # not cost the line — rank 0 then prints it with the error in "gather" and every rank leaves.
        printed = threading.Lock()

        def give_up(info=None):
            if printed.acquire(blocking=False):
                if rank == 0:
                    out["gather"] = info or {"error": f"no result within {args.gather_timeout:.0f} s", "verified": False}
                    print(json.dumps(out), flush=True)
                # the measured line is out; the run still FAILED — a driver that reads only the status must see it.  (os._exit:
                # this is the watchdog's thread, or a rank whose peers sit inside a collective that will not return)
                os._exit(3)

        dog = threading.Timer(args.gather_timeout, give_up)
        dog.daemon = True
        dog.start()
        try:
            info = gather_leg(ctx, dist, outs[3] if method == "dqm" else outs[-1], rank, world, C, T, barrier)
        except Exception as e:  # noqa: BLE001 - reported in the line
            info = {"error": f"{type(e).__name__}: {e}"[:300], "verified": False}
            if world > 1:  # the other ranks may be inside the collective: leave (their watchdogs end them)
                give_up(info)
        if not printed.acquire(blocking=False):
            time.sleep(3600)  # the watchdog is printing
        dog.cancel()
        if rank == 0:
            out["gather"] = info
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def gather_leg(ctx, dist, scen_dev, rank, world, C, T, barrier):
    """Native gather of the per-rank scen blocks to rank 0 (xsdba_hip_gather: RCCL sends / receives straight from the device
    buffers on the library's stream).  Self-check: rank 0 compares rows of every rank's block of the gathered cube with what
    that rank holds.  Returns the "gather" object of the line on rank 0."""
    from xsdba_amd._lib import DeviceArray
    from xsdba_amd.sharding import comm_info, gather_device, init_comm, shard_plan

    # (RCCL prints a version banner on stdout when its first communicator comes up: stdout belongs to the JSON line)
    sys.stdout.flush()
    saved_fd = os.dup(1)
    os.dup2(2, 1)
    try:
        init_comm(ctx, rank, world)
    finally:
        sys.stdout.flush()
        os.dup2(saved_fd, 1)
        os.close(saved_fd)
    r_rccl, n_rccl = comm_info(ctx)
    assert (r_rccl, n_rccl) == (rank, world), (r_rccl, n_rccl)
    assert shard_plan(C * world, world, rank) == (rank * C, C)
    full = gather_device(ctx, scen_dev, C * world, root=0)  # warm-up (connection set-up) + the buffer
    barrier()
    g0 = time.perf_counter()
    gather_device(ctx, scen_dev, C * world, root=0, out=full)
    ctx.synchronize()
    gather_ms = (time.perf_counter() - g0) * 1e3
    probe_rows = [0, C // 2, C - 1]

    def rows_of(dev, rows):  # a few rows of a [cells, T] float32 device array (the gathered cube is tens of GB: never all of it)
        return np.stack([DeviceArray.from_ptr(ctx, dev.ptr + r * T * 4, (T,), np.float32).download() for r in rows])

    mine = rows_of(scen_dev, probe_rows)
    if dist is not None:
        box = [None] * world if rank == 0 else None
        dist.gather_object(mine, box, dst=0)
    else:
        box = [mine]
    if rank != 0:
        return None
    ok = all(np.array_equal(rows_of(full, [r * C + i for i in probe_rows]), rows, equal_nan=True) for r, rows in enumerate(box))
    return {"ms": round(gather_ms, 3), "ranks_seen_by_rccl": n_rccl, "bytes": int(C * world * T * 4),
            "GB/s": round(C * (world - 1) * T * 4 / (gather_ms * 1e-3) / 1e9, 1), "verified": bool(ok)}


def spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: run `torch.distributed.run --nproc-per-node N bench.py <same arguments>`
    as a child process, relay rank 0's JSON line and return the child's status."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(Path(__file__).resolve()), *sys.argv[1:]]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes needs it on this driver
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout.splitlines():
        if ln.startswith('{"metric"'):
            line = ln
        elif ln.strip():
            print(ln, file=sys.stderr)
    if line is None:
        print(f"bench.py: the {n} ranks printed no result line (status {proc.returncode})", file=sys.stderr)
        return proc.returncode or 1
    print(line, flush=True)
    return proc.returncode


def dry_run(world: int, rank: int, cells: int) -> None:
    """Rendezvous only (gloo), no GPU work: what tests/test_bench_spawn.py runs on a box without a GPU."""
    seen = 1
    if world > 1:
        import torch
        import torch.distributed as dist

        dist.init_process_group("gloo")
        t = torch.ones(1, dtype=torch.int64)
        dist.all_reduce(t)
        seen = int(t.item())
    if rank == 0:
        print(json.dumps({"metric": "dry-run", "n_gpus": world, "ranks_seen": seen, "cells_per_gpu": cells, "dry_run": True}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def api_level(ctx, ref, hist, sim, method, group, W, kind, Q, C, T, rank_window):
    """Secondary figure, never `value`: the same workload through the drop-in API — <Class>.train(ref, hist, group=...,
    nquantiles=...) -> .adjust(sim) on labelled arrays (reference adjustment.py:227-316).
      device_inputs: DataArrays wrapping the device cubes of the timed region (tables and scen stay on the device);
      host_inputs:   numpy-backed DataArrays of a bounded sample of the same cells (uploaded through page-locked blocks,
                     scen downloaded: the PCIe-inclusive rate)."""
    import warnings

    import xsdba_amd as xs
    from xsdba_amd.labeled import DataArray

    cls = {"eqm": xs.EmpiricalQuantileMapping, "dqm": xs.DetrendedQuantileMapping, "qdm": xs.QuantileDeltaMapping}[method]
    grp = xs.Grouper("time" if group == "time" else f"time.{group}", window=W) if group != "time" else "time"
    times = xs.date_range("2000-01-01", T, "noleap").values
    units = "K" if kind == "+" else "mm/d"
    adj_kw = {"rank_window": rank_window} if method == "qdm" else {}

    def wrap(a, n):
        return DataArray(a, ("cell", "time"), {"cell": np.arange(n), "time": times}, {"units": units})

    def run(r, h, s_):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            obj = cls.train(r, h, group=grp, nquantiles=Q, kind=kind)
            return obj.adjust(s_, **adj_kw)

    res = {}
    dev = [wrap(a, C) for a in (ref, hist, sim)]
    from xsdba_amd._lib import default_context

    api_ctx = default_context()  # the context the mirror launches on (not the one the cubes were generated on)
    # warm-up: two passes whose results are alive together, as in the timed loop (`scen = run(...)` holds the previous result
    # while the next is produced): every buffer size the loop asks for is then in the context's pool twice
    w1 = run(*dev)
    w2 = run(*dev)
    del w1, w2
    ctx.synchronize()
    api_ctx.synchronize()
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        scen = run(*dev)
    api_ctx.synchronize()
    res["device_inputs_cells_per_s"] = round(C * reps / (time.perf_counter() - t0), 1)
    del scen
    n = min(C, 16384)
    host = [wrap(first_rows(ctx, a, n, T), n) for a in (ref, hist, sim)]
    saved = os.environ.get("XSDBA_AMD_DEVICES")
    os.environ["XSDBA_AMD_DEVICES"] = str(api_ctx.device)  # ONE context first (the figure of rounds 1-4)
    run(*host).values  # warm-up
    t0 = time.perf_counter()
    out = run(*host).values  # the download is part of it
    dt_h = time.perf_counter() - t0
    res["host_inputs_cells_per_s"] = round(n / dt_h, 1)
    res["host_inputs_sample_cells"] = n
    res["host_inputs_GBps"] = round(4 * n * T * 4 / dt_h / 1e9, 2)  # three series up, one down
    assert out.shape == (n, T)
    # the node behind the API: the same host cubes with their cell blocks dealt over one context per device (map_blocks,
    # XSDBA_AMD_DEVICES; what dask's workers are to the reference).  On a one-GPU box: two contexts on that device — a block's
    # upload then runs under the other context's kernels.
    from xsdba_amd._lib import device_count, worker_contexts

    try:
        nd = device_count()
        spec = "all" if nd > 1 else "0,0"
        os.environ["XSDBA_AMD_DEVICES"] = spec
        nctx = len(worker_contexts())
        ref_out = out
        run(*host).values  # warm-up (the other contexts' plans, pools and scratch)
        t0 = time.perf_counter()
        out = run(*host).values
        dt_m = time.perf_counter() - t0
        res["host_inputs_multi"] = {"contexts": nctx, "devices": spec, "cells_per_s": round(n / dt_m, 1),
                                    "GBps": round(4 * n * T * 4 / dt_m / 1e9, 2), "equal_to_one_context": bool(np.array_equal(out, ref_out, equal_nan=True))}
    finally:
        if saved is None:
            os.environ.pop("XSDBA_AMD_DEVICES", None)
        else:
            os.environ["XSDBA_AMD_DEVICES"] = saved
    return res


def first_rows(ctx, dev, n, T):
    """the first `n` series of a cell-major [cells, T] float32 device cube on the host: a view of those rows is downloaded,
    never the cube (5.7 GB at the default size)"""
    from xsdba_amd._lib import DeviceArray

    return DeviceArray.from_ptr(ctx, dev.ptr, (n, T), np.float32).download().reshape(n, T)


def pmc_traffic(workload, cells, days, kernel):
    """(HBM bytes per launch of `kernel`, where they come from).  The bytes are NOT measured by this run: they are the
    rocprofv3 PMC passes of this very configuration committed under profiles/ (*traffic*.json: separate --pmc FETCH_SIZE /
    WRITE_SIZE runs of this script, gfx950 corrections applied), and `traffic_source` names the file; (None, reason) when
    no committed PMC run matches the workload, cells and days being benchmarked."""
    # the newest set first: "<round>_final_*", then the versioned sets in descending order (v10 before v9)
    def age(f):
        # "rNN_final*" is the round's last set, "rNN_vK*" an intermediate one, a plain "rNN_*" the first
        m = re.search(r"r(\d+)_", f.name)
        v = re.search(r"r\d+_v(\d+)", f.name)
        return (int(m.group(1)), 10**6 if re.search(r"r\d+_final", f.name) else (int(v.group(1)) if v else 0)) if m else (-1, -1)

    for f in sorted((ROOT / "profiles").glob("*traffic*.json"), key=age, reverse=True):
        try:
            d = json.loads(f.read_text())
        except Exception:
            continue
        if d.get("workload") == workload and d.get("cells") == cells and d.get("days") == days:
            k = d.get("kernels", {}).get(kernel.split("(")[0])
            if k:
                return k["traffic_bytes"], f"profiles/{f.name} (committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this configuration; not measured in this run)"
    return None, f"none: no committed PMC pass under profiles/ for workload={workload}, cells={cells}, days={days}, kernel={kernel}"


def cpu_baseline(ctx, hip, ref, hist, sim, method, gidx, G, W, q, kind, tnum, C, T, target_s, time_major, rank_window=False,
                 synth_args=None):
    """The CPU oracle (a restatement of the reference algorithm, kind="port") timed on this box's host cores
    on a bounded sample of the SAME device-generated cells; reported beside the GPU number, never the target."""
    import oracle

    def fetch(which, dev, n):
        # only the rows the sample uses ever cross PCIe: a view of the first n series of a cell-major cube; for a time-major
        # cube (whose series are columns) the same n cells regenerated cell-major by the counter-based generator
        if not time_major:
            return first_rows(ctx, dev, n, T)
        tmp = ctx.synth((n, T), np.float32, *synth_args[:1], which, *synth_args[1:], time_major=False)
        try:
            return tmp.download().reshape(n, T)
        finally:
            tmp.free()

    def run(r, h, s):
        t0 = time.perf_counter()
        if method == "dqm":
            af, hq, sc = oracle.dqm_train(r, h, gidx, G, W, q, kind)
            oracle.dqm_adjust(s, af, hq, sc, gidx, W, tnum, 1, "nearest", "constant", kind)
        else:
            af, hq = oracle.eqm_train(r, h, gidx, G, W, q, kind)
            if method == "qdm":
                oracle.qdm_adjust(s, af, q, gidx, W, rank_window, "nearest", "constant", kind)
            else:
                oracle.qm_adjust(s, af, hq, gidx, "nearest", "constant", kind)
        return time.perf_counter() - t0

    nmax = min(C, 8192)
    r, h, s = fetch(0, ref, nmax), fetch(1, hist, nmax), fetch(2, sim, nmax)
    # thread scaling first: the box may expose far more logical CPUs than this job's share of cores, and OpenMP over all of
    # them then measures oversubscription, not the algorithm.  Short probes (a few cells per thread) at 1, 4, 16, ... threads;
    # the count with the best rate runs the bounded sample.
    ncpu = os.cpu_count() or 1
    try:
        ncpu = min(ncpu, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    cands = sorted({c for c in (1, 4, 16, 32, 64, 128, 256, ncpu) if c <= ncpu})
    scaling = {}
    for nt in cands:
        oracle.set_num_threads(nt)
        # (the oracle deals cells in chunks of four, schedule(dynamic, 4): eight cells per thread keep every thread busy — two
        #  per thread left half of them idle and under-reported the multi-thread rates)
        probe = min(nmax, max(8 * nt, 8))
        tp = run(r[:probe], h[:probe], s[:probe])
        scaling[nt] = probe / tp
        if tp > 4.0:  # a probe this slow will not be beaten by more threads on the same cores
            break
    threads = max(scaling, key=scaling.get)
    oracle.set_num_threads(threads)
    n = int(min(nmax, max(8 * threads, scaling[threads] * target_s)))
    n = max(threads, (n // (4 * threads)) * 4 * threads)
    tt = run(r[:n], h[:n], s[:n])
    table = ", ".join(f"{k}: {v:.1f}" for k, v in scaling.items())
    return {
        "value": round(n / tt, 2), "unit": "grid-cells/s", "cores": threads, "kind": "port",
        "per_thread": round(n / tt / threads, 2), "thread_scaling_cells_per_s": {str(k): round(v, 1) for k, v in scaling.items()},
        # the box hands a job a SHARE of its cores (a cgroup quota): threads beyond it time-slice.  value / the one-thread rate
        # says how many cores' worth of work the sample actually got
        "cores_effective": round((n / tt) / scaling[1], 1) if scaling.get(1) else None,
        "sample": f"{n} of the {C} cells of rank 0 (same device-generated data), full train+adjust, {tt:.1f} s wall, OpenMP over "
                  f"cells on {threads} threads ({n / tt / threads:.1f} cells/s per thread; probes at [threads: cells/s] {table}; "
                  f"{os.cpu_count()} logical host CPUs visible); CPU restatement of the reference algorithm "
                  "(oracle/xsdba_oracle.c), not the reference's dask/numba stack",
    }


if __name__ == "__main__":
    main()
