#!/usr/bin/env python3
"""k_lin2d_adjust with extrapolation="constant" vs "nan" (the latter skips the group-interpolated bounds): what the four
np_interp_at evaluations per element cost."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import group_index  # noqa: E402
from xsdba_amd import hip  # noqa: E402
ctx = hip.Context(0)
C, T, Q = 64800, 10950, 20
gidx, G = group_index("month", T)
q = ((np.arange(Q) + 0.5) / Q).astype(np.float32)
ref, hist, sim = (ctx.synth((C, T), np.float32, 0, i, 1) for i in range(3))
af, hq = hip.eqm_train(ref, hist, gidx, G, 1, q, "+", ctx=ctx)
d0 = np.arange(T) % 365
cum = np.concatenate([[0], np.cumsum([31, 28, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31])])
m = np.searchsorted(cum, d0, side="right") - 1
newg = (m + 1) - 0.5 + (d0 - cum[m] + 1) / np.diff(cum)[m]
gcoord = np.arange(0, G + 2.0)
for ex in ("constant", "nan"):
    fn = lambda: hip.qm_adjust_linear2d(sim, af, hq, gcoord, newg, ex, "+", ctx=ctx)  # noqa: E731
    fn(); ctx.synchronize()
    ctx.profile_enable(True); ctx.profile_read()
    for _ in range(3):
        fn()
    ctx.synchronize()
    print(ex, {k: round(v["total_ms"] / 3, 2) for k, v in ctx.profile_read().items() if v["count"]})
    ctx.profile_enable(False)
