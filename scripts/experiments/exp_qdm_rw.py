import sys, numpy as np
sys.path.insert(0, ".")
from bench import group_index
from xsdba_amd import hip
ctx = hip.default_context(0)
C, T, Q = 64800, 10950, 20
q = ((np.arange(Q) + 0.5) / Q).astype(np.float32)
sim = ctx.synth((C, T), np.float32, 1, 2, 1, 0)
gidx, G = group_index("dayofyear", T)
af = ctx.to_device((1.0 + 0.1 * np.random.default_rng(0).standard_normal((C, G, Q))).astype(np.float32))
for _ in range(2):
    ctx.profile_enable(True); ctx.profile_read(reset=True)
    s2, _ = hip.qdm_adjust(sim, af, q, gidx, 31, True, "nearest", "constant", "*", want_sim_q=False, ctx=ctx)
    ctx.synchronize()
    r = {k: round(v["total_ms"], 2) for k, v in ctx.profile_read(reset=True).items() if v["count"] and v["total_ms"] > 0.05}
    s2.free()
print("doy31_rw", r)
