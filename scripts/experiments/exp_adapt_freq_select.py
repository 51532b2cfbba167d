"""adapt_freq: the LDS sorts (XSDBA_HIP_AF_SELECT=0) against the selection (the default) on rows where both apply, per-kernel times from the
library's own profile; 8 192 cells x 10 950 days, sim drier than ref everywhere (every block adapts).
    python scripts/experiments/exp_adapt_freq_select.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from xsdba_amd import hip  # noqa: E402

C, T = 8192, 10950
rng = np.random.default_rng(0)
ref = rng.gamma(0.8, 4.0, (C, T)).astype(np.float32)
sim = rng.gamma(0.8, 4.0, (C, T)).astype(np.float32)
ref[rng.random((C, T)) < 0.3] = 0.0
sim[rng.random((C, T)) < 0.5] = 0.0
d0 = np.arange(T) % 365
cum = np.concatenate([[0], np.cumsum([31, 28, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31])])
mon = (np.searchsorted(cum, d0, side="right") - 1).astype(np.int32)
ctx = hip.default_context()
dsim, dref = ctx.to_device(sim), ctx.to_device(ref)
for gname, gidx, G, W in (("time", np.zeros(T, np.int32), 1, 1), ("month", mon, 12, 1), ("doy31", d0.astype(np.int32), 365, 31)):
    for sel in (False, True, False, True):
        if sel:
            os.environ.pop("XSDBA_HIP_AF_SELECT", None)
        else:
            os.environ["XSDBA_HIP_AF_SELECT"] = "0"
        hip.adapt_freq(dsim, dref, gidx, G, W, 1.0, seed=1)
        ctx.synchronize()
        ctx.profile_enable(True); ctx.profile_read(reset=True)
        t0 = time.perf_counter()
        hip.adapt_freq(dsim, dref, gidx, G, W, 1.0, seed=1)
        ctx.synchronize()
        wall = 1e3 * (time.perf_counter() - t0)
        kern = {k: round(v["total_ms"], 2) for k, v in ctx.profile_read(reset=True).items() if v["count"] and v["total_ms"] > 0.02}
        ctx.profile_enable(False)
        print(f"{gname:6s} select={int(sel)}  {wall:8.2f} ms per {C} cells  {kern}", flush=True)
os.environ.pop("XSDBA_HIP_AF_SELECT", None)
