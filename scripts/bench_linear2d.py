#!/usr/bin/env python3
"""Time the grouped interp="linear" adjustment (k_lin2d_build + k_lin2d_adjust) next to the "nearest" one on the cfg2-like
cube: 64800 cells x 10950 days, monthly groups, 20 quantiles, everything resident in HBM.
    python scripts/bench_linear2d.py [--cells N] [--group month|dayofyear]"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import group_index  # noqa: E402
from xsdba_amd import hip  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--cells", type=int, default=64800)
ap.add_argument("--days", type=int, default=10950)
ap.add_argument("--group", default="month")
ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
ctx = hip.Context(0)
C, T, Q = a.cells, a.days, 20
W = 31 if a.group == "dayofyear" else 1
gidx, G = group_index(a.group, T)
q = ((np.arange(Q) + 0.5) / Q).astype(np.float32)
ref, hist, sim = (ctx.synth((C, T), np.float32, 0, i, 1) for i in range(3))
af, hq = hip.eqm_train(ref, hist, gidx, G, W, q, "+", ctx=ctx)
d0 = np.arange(T) % 365
if a.group == "month":
    cum = np.concatenate([[0], np.cumsum([31, 28, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31])])
    m = np.searchsorted(cum, d0, side="right") - 1
    newg = (m + 1) - 0.5 + (d0 - cum[m] + 1) / np.diff(cum)[m]
else:
    newg = d0 + 1.0
gcoord = np.arange(0, G + 2.0)
res = {}
for name, fn in (("nearest", lambda: hip.qm_adjust(sim, af, hq, gidx, "nearest", "constant", "+", ctx=ctx)),
                 ("linear", lambda: hip.qm_adjust_linear2d(sim, af, hq, gcoord, newg, "constant", "+", ctx=ctx)),
                 ("cubic", lambda: hip.qm_adjust_linear2d(sim, af, hq, gcoord, newg, "constant", "+", ctx=ctx, cubic=True))):
    fn(); ctx.synchronize()
    ctx.profile_enable(True); ctx.profile_read()
    t0 = time.perf_counter()
    for _ in range(a.reps):
        fn()
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / a.reps
    prof = {k: round(v["total_ms"] / a.reps, 3) for k, v in ctx.profile_read().items()}
    ctx.profile_enable(False)
    res[name] = {"ms": round(dt * 1e3, 2), "cells_per_s": round(C / dt), "kernels_ms": prof}
print(json.dumps({"cells": C, "days": T, "group": a.group, **res}))
