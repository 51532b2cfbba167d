"""More settings: monthly groups with multiplicative precipitation-like data, DQM '*', QDM '+', 1-D interpolations."""
import sys, time, numpy as np
sys.path.insert(0, ".")
from bench import group_index
from xsdba_amd import hip
C = int(sys.argv[1]) if len(sys.argv) > 1 else 16200
ctx = hip.default_context(0)
T, Q = 10950, 20
q = ((np.arange(Q) + 0.5) / Q).astype(np.float32)
tnum = (np.arange(T, dtype=np.float64) + 10950.0) * 86400e9
def timed(label, fn):
    res = None
    for _ in range(2):
        ctx.profile_enable(True); ctx.profile_read(reset=True)
        t0 = time.perf_counter(); out = fn(); ctx.synchronize(); dt = time.perf_counter() - t0
        res = (C / dt, {k: round(v["total_ms"], 2) for k, v in ctx.profile_read(reset=True).items() if v["count"] and v["total_ms"] > 0.05})
        for a in out:
            if hasattr(a, "free"): a.free()
    print(f"{label:44s} {res[0]/1e6:6.2f} M cells/s  {res[1]}", flush=True)
for gen, kind in ((0, "+"), (1, "*")):
    ref, hist, sim = (ctx.synth((C, T), np.float32, gen, i, 1, 0) for i in range(3))
    for gname in ("month", "dayofyear", "time"):
        gidx, G = group_index(gname, T)
        W = 31 if gname == "dayofyear" else 1
        def eqm(interp="nearest", extrap="constant"):
            af, hq = hip.eqm_train(ref, hist, gidx, G, W, q, kind, ctx=ctx)
            return af, hq, hip.qm_adjust(sim, af, hq, gidx, interp, extrap, kind, ctx=ctx)
        timed(f"EQM {gname} {kind} nearest", eqm)
        if gname == "time":
            timed(f"EQM {gname} {kind} linear", lambda: eqm("linear"))
            timed(f"EQM {gname} {kind} cubic/nan", lambda: eqm("cubic", "nan"))
        def dqm():
            af, hq, sc = hip.dqm_train(ref, hist, gidx, G, W, q, kind, ctx=ctx)
            scen, trend = hip.dqm_adjust(sim, af, hq, sc, gidx, W, tnum, 1, "nearest", "constant", kind, ctx=ctx)
            return af, hq, sc, scen, trend
        timed(f"DQM {gname} {kind}", dqm)
        def qdm(rw):
            af, hq = hip.eqm_train(ref, hist, gidx, G, W, q, kind, ctx=ctx)
            scen, _ = hip.qdm_adjust(sim, af, q, gidx, W, rw, "nearest", "constant", kind, want_sim_q=False, ctx=ctx)
            return af, hq, scen
        timed(f"QDM {gname} {kind}", lambda: qdm(False))
        if gname == "dayofyear": timed(f"QDM {gname} {kind} rank_window", lambda: qdm(True))
    for a in (ref, hist, sim): a.free()
