import sys, numpy as np
sys.path.insert(0, ".")
from bench import group_index
from xsdba_amd import hip
ctx = hip.default_context(0)
C, T, Q = 64800, 10950, 20
q = ((np.arange(Q) + 0.5) / Q).astype(np.float32)
sim = ctx.synth((C, T), np.float32, 0, 2, 1, 0)
ref = ctx.synth((C, T), np.float32, 0, 0, 1, 0); hist = ctx.synth((C, T), np.float32, 0, 1, 1, 0)
gidx, G = group_index("dayofyear", T)
af, hq = hip.eqm_train(ref, hist, gidx, G, 31, q, "+", ctx=ctx)
for _ in range(3):
    ctx.profile_enable(True); ctx.profile_read(reset=True)
    s2 = hip.qm_adjust(sim, af, hq, gidx, "nearest", "constant", "+", ctx=ctx)
    ctx.synchronize()
    r = {k: round(v["total_ms"], 3) for k, v in ctx.profile_read(reset=True).items() if v["count"] and v["total_ms"] > 0.05}
    s2.free()
print("adjust", r)
