import sys, numpy as np
sys.path.insert(0, ".")
from bench import group_index
from xsdba_amd import hip
ctx = hip.default_context(0)
C, T, Q = 64800, 10950, 20
q = ((np.arange(Q) + 0.5) / Q).astype(np.float32)
ref = ctx.synth((C, T), np.float32, 0, 0, 1, 0); hist = ctx.synth((C, T), np.float32, 0, 1, 1, 0)
gidx, G = group_index("dayofyear", T)
for _ in range(3):
    ctx.profile_enable(True); ctx.profile_read(reset=True)
    out = hip.dqm_train(ref, hist, gidx, G, 31, q, "+", ctx=ctx)
    ctx.synchronize()
    r = {k: round(v["total_ms"], 3) for k, v in ctx.profile_read(reset=True).items() if v["count"] and v["total_ms"] > 0.05}
    for o in out: o.free()
print("dqm_train", r)
