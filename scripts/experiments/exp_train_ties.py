"""The training kernels against ties in the data: continuous values, values rounded to 0.01 / 0.1 / 1, zero-inflated.  16 384 cells."""
import sys, numpy as np
sys.path.insert(0, ".")
from bench import group_index
from xsdba_amd import hip
ctx = hip.default_context(0)
C, T, Q = 16384, 10950, 20
q = ((np.arange(Q) + 0.5) / Q).astype(np.float32)
rng = np.random.default_rng(0)
for gen, kind in ((0, "+"), (1, "*")):
    ref0 = np.asarray(ctx.synth((C, T), np.float32, gen, 0, 1, 0).download()); hist0 = np.asarray(ctx.synth((C, T), np.float32, gen, 1, 1, 0).download())
    def variants():
        yield "continuous", ref0, hist0
        for d in (2, 1, 0):
            yield f"rounded {10.0 ** -d:g}", np.round(ref0, d), np.round(hist0, d)
        if gen == 1:
            a, b = ref0.copy(), hist0.copy()
            a[rng.random(a.shape) < 0.5] = 0.0; b[rng.random(b.shape) < 0.45] = 0.0
            yield "zero-inflated", a, b
    for gname, W in (("dayofyear", 31), ("month", 1), ("time", 1)):
        gidx, G = group_index(gname, T)
        for name, a, b in variants():
            ref, hist = ctx.to_device(a), ctx.to_device(b)
            for _ in range(2):
                ctx.profile_enable(True); ctx.profile_read(reset=True)
                af, hq = hip.eqm_train(ref, hist, gidx, G, W, q, kind, ctx=ctx)
                ctx.synchronize()
                r = {k: round(v["total_ms"], 2) for k, v in ctx.profile_read(reset=True).items() if v["count"] and v["total_ms"] > 0.02}
                af.free(); hq.free()
            print(f"gen{gen} {gname:9s} {name:14s}", r)
            ref.free(); hist.free()
