"""adapt_freq / jitter kernel timings on precipitation-like cubes (timing aid)."""
import sys, time, numpy as np
sys.path.insert(0, ".")
from xsdba_amd import hip
C, T = int(sys.argv[1]) if len(sys.argv) > 1 else 16384, 10950
G, W = 365, 31
ctx = hip.default_context(0)
doy = (np.arange(T) % 365).astype(np.int32)
ref, hist = (ctx.synth((C, T), np.float32, 1, i, 3, 0) for i in range(2))
ctx.synchronize()
for _ in range(2):
    ctx.profile_enable(True); ctx.profile_read(reset=True)
    t0 = time.perf_counter()
    hj = hip.jitter(hist, lower=0.5, seed=1, ctx=ctx)
    out = hip.adapt_freq(hj, ref, doy, G, W, 8.0, seed=2, ctx=ctx)
    ctx.synchronize()
    dt = time.perf_counter() - t0
    print(C / dt, "cells/s", {k: round(v["total_ms"], 2) for k, v in ctx.profile_read(reset=True).items() if v["count"]},
          "dP0 mean", float(np.nanmean(out[2])))
