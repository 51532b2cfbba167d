"""Train + adjust timings over the options a user is likely to vary (window, nodes, series length, kind) — looks for settings that
fall off the fast kernels.   python scripts/experiments/sweep_params.py [cells]"""
import sys, time, numpy as np
sys.path.insert(0, ".")
from xsdba_amd import hip
C = int(sys.argv[1]) if len(sys.argv) > 1 else 16200
ctx = hip.default_context(0)
def run(T, W, Q, kind, gen):
    G = 365
    doy = (np.arange(T) % 365).astype(np.int32)
    q = ((np.arange(Q) + 0.5) / Q).astype(np.float32)
    ref, hist, sim = (ctx.synth((C, T), np.float32, gen, i, 1, 0) for i in range(3))
    ctx.synchronize()
    res = None
    for _ in range(2):
        ctx.profile_enable(True); ctx.profile_read(reset=True)
        t0 = time.perf_counter()
        af, hq = hip.eqm_train(ref, hist, doy, G, W, q, kind, ctx=ctx)
        scen = hip.qm_adjust(sim, af, hq, doy, "nearest", "constant", kind, ctx=ctx)
        ctx.synchronize()
        dt = time.perf_counter() - t0
        res = (C / dt, {k: round(v["total_ms"], 2) for k, v in ctx.profile_read(reset=True).items() if v["count"] and v["total_ms"] > 0.05})
        for a in (af, hq, scen):
            if hasattr(a, "free"): a.free()
    for a in (ref, hist, sim): a.free()
    print(f"T={T:6d} W={W:3d} Q={Q:3d} kind={kind} gen={gen}: {res[0]/1e6:6.2f} M cells/s  {res[1]}", flush=True)
for W in (1, 15, 31, 61, 91):
    run(10950, W, 20, "+", 0)
for T in (3650, 7300, 10950, 14600, 21900):
    run(T, 31, 20, "+", 0)
run(10950, 31, 20, "*", 1)
run(10950, 31, 50, "*", 1)
