import sys, time, numpy as np
sys.path.insert(0, ".")
from xsdba_amd import hip
ctx = hip.default_context(0)
T, Q, G, W = 10950, 20, 365, 31
doy = (np.arange(T) % 365).astype(np.int32)
q = ((np.arange(Q) + 0.5) / Q).astype(np.float32)
for C in (1, 64, 1024, 8192):
    ref, hist, sim = (ctx.synth((C, T), np.float32, 0, i, 1, 0) for i in range(3))
    ctx.synchronize()
    for rep in range(3):
        t0 = time.perf_counter()
        af, hq = hip.eqm_train(ref, hist, doy, G, W, q, "+", ctx=ctx)
        scen = hip.qm_adjust(sim, af, hq, doy, "nearest", "constant", "+", ctx=ctx)
        ctx.synchronize()
        dt = time.perf_counter() - t0
        for a in (af, hq, scen): a.free()
    print(C, "cells:", round(dt * 1e3, 3), "ms per train+adjust call")
