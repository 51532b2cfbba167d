"""k_qdm_cell against ties in the data (doy/31 rank_window and time.month): continuous values, values rounded to 0.1 / 0.01
(measured temperature, precipitation), zero-inflated precipitation.  16 384 cells."""
import sys, numpy as np
sys.path.insert(0, ".")
from bench import group_index
from xsdba_amd import hip
ctx = hip.default_context(0)
C, T, Q = 16384, 10950, 20
q = ((np.arange(Q) + 0.5) / Q).astype(np.float32)
base = np.asarray(ctx.synth((C, T), np.float32, 1, 2, 1, 0).download())
rng = np.random.default_rng(0)
cases = {"continuous": base, "rounded 0.01": np.round(base, 2), "rounded 0.1": np.round(base, 1)}
z = base.copy(); z[rng.random(z.shape) < 0.5] = 0.0
cases["zero-inflated"] = z
for gname, W, rw in (("dayofyear", 31, True), ("month", 1, False)):
    gidx, G = group_index(gname, T)
    af = ctx.to_device((1.0 + 0.1 * rng.standard_normal((C, G, Q))).astype(np.float32))
    for name, arr in cases.items():
        sim = ctx.to_device(arr)
        for _ in range(2):
            ctx.profile_enable(True); ctx.profile_read(reset=True)
            s2, _ = hip.qdm_adjust(sim, af, q, gidx, W, rw, "nearest", "constant", "*", want_sim_q=False, ctx=ctx)
            ctx.synchronize()
            r = {k: round(v["total_ms"], 2) for k, v in ctx.profile_read(reset=True).items() if v["count"] and v["total_ms"] > 0.05}
            s2.free()
        print(gname, name, r)
        sim.free()
    af.free()
