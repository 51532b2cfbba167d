import sys
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import numpy as np
from xsdba_amd import hip
import test_adjust_time_major_gpu as t
ctx = hip.default_context(0)
for gname, Q, kind, extrap, spread in [("month", 48, "+", "constant", 1.0), ("doy", 20, "+", "constant", 12.0), ("month", 21, "*", "constant", 8.0), ("doy", 32, "+", "nan", 6.0)]:
    for seed in range(12):
        rng = np.random.default_rng(seed)
        C, T = 131, 3 * 365 + 17
        gidx, G = t._groups(gname, T)
        hq, af = t._tables(rng, C, G, Q, kind, spread)
        sim = (hq[:, gidx, :].mean(-1) + (3.0 if kind == "+" else 1.0) * spread * rng.standard_normal((C, T))).astype(np.float32)
        sim[rng.random(sim.shape) < 0.01] = np.nan
        sim[5] = np.nan
        hq[7] = np.nan; af[7] = np.nan
        af[9, 3, 2] = np.nan
        hq[11, G // 2, :] = np.nan
        hq[13, :, Q - 1] = np.nan
        want = hip.qm_adjust(sim, af, hq, gidx, "nearest", extrap, kind)
        got = hip.qm_adjust(t._tm(ctx, sim), ctx.to_device(af), ctx.to_device(hq), gidx, "nearest", extrap, kind, ctx=ctx, out_time_major=True).download().reshape(T, C).T
        bad = ~((got == want) | (np.isnan(got) & np.isnan(want)))
        if bad.any():
            cells = np.unique(np.argwhere(bad)[:, 0])
            c, tt = np.argwhere(bad)[0]
            print(gname, Q, kind, "seed", seed, "bad", int(bad.sum()), "cells", cells[:10], "first", c, tt, "g", gidx[tt], "x", sim[c, tt], got[c, tt], want[c, tt])
        else:
            print(gname, Q, kind, "seed", seed, "ok")
