"""EQM doy/31 train+adjust on float64 cubes (generic kernels) — timing aid."""
import sys, time, numpy as np
sys.path.insert(0, ".")
from xsdba_amd import hip
C, T = int(sys.argv[1]) if len(sys.argv) > 1 else 8192, 10950
G, W, Q = 365, 31, 20
ctx = hip.default_context(0)
doy = (np.arange(T) % 365).astype(np.int32)
q = ((np.arange(Q) + 0.5) / Q)
ref, hist, sim = (ctx.synth((C, T), np.float64, 0, i, 1, 0) for i in range(3))
ctx.synchronize()
for _ in range(2):
    ctx.profile_enable(True); ctx.profile_read(reset=True)
    t0 = time.perf_counter()
    af, hq = hip.eqm_train(ref, hist, doy, G, W, q.astype(np.float64), "+", ctx=ctx)
    scen = hip.qm_adjust(sim, af, hq, doy, "nearest", "constant", "+", ctx=ctx)
    ctx.synchronize()
    dt = time.perf_counter() - t0
    print(C / dt, "cells/s", {k: round(v["total_ms"], 2) for k, v in ctx.profile_read(reset=True).items() if v["count"]})
# QDM adjust on the same float64 cube: k_qdm_lanes<double> against the generic kernel (XSDBA_HIP_QDM=count)
import os
afq = ctx.to_device((1.0 + 0.1 * np.random.default_rng(0).standard_normal((C, G, Q))))
for force in (None, "count"):
    if force: os.environ["XSDBA_HIP_QDM"] = force
    else: os.environ.pop("XSDBA_HIP_QDM", None)
    for _ in range(2):
        ctx.profile_enable(True); ctx.profile_read(reset=True)
        s2, _ = hip.qdm_adjust(sim, afq, q.astype(np.float64), doy, W, False, "nearest", "constant", "*", want_sim_q=False, ctx=ctx)
        ctx.synchronize()
        print("qdm float64", force or "lanes", {k: round(v["total_ms"], 2) for k, v in ctx.profile_read(reset=True).items() if v["count"]})
os.environ.pop("XSDBA_HIP_QDM", None)
