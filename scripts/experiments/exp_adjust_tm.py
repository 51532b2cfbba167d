"""k_qm_adjust_tm (time-major sim in, time-major scen out) against k_qm_adjust on the bench's EQM doy/31 tables: per-kernel ms."""
import sys
import numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from xsdba_amd import hip

C = int(sys.argv[1]) if len(sys.argv) > 1 else 64800
T, Q, G, W = 10950, 20, 365, 31
ctx = hip.Context(0)
gidx = (np.arange(T) % 365).astype(np.int32)
q = ((np.arange(Q) + 0.5) / Q).astype(np.float32)
ref, hist, sim = (ctx.synth((C, T), np.float32, 0, i, 1) for i in range(3))
sim_tm = ctx.synth((C, T), np.float32, 0, 2, 1, time_major=True)
af, hq = hip.eqm_train(ref, hist, gidx, G, W, q, "+", ctx=ctx)
o_cm, o_tm = ctx.empty((C, T), np.float32), ctx.empty((T, C), np.float32)
for _ in range(2):
    hip.qm_adjust(sim, af, hq, gidx, "nearest", "constant", "+", ctx=ctx, out=o_cm)
    hip.qm_adjust(sim_tm, af, hq, gidx, "nearest", "constant", "+", ctx=ctx, out=o_tm, out_time_major=True)
ctx.synchronize()
ctx.profile_enable(True)
ctx.profile_read(reset=True)
for _ in range(5):
    hip.qm_adjust(sim, af, hq, gidx, "nearest", "constant", "+", ctx=ctx, out=o_cm)
    hip.qm_adjust(sim_tm, af, hq, gidx, "nearest", "constant", "+", ctx=ctx, out=o_tm, out_time_major=True)
ctx.synchronize()
for k, v in ctx.profile_read(reset=True).items():
    if v["count"]:
        print(k, v["count"], "%.3f ms" % (v["total_ms"] / v["count"]))
a = o_cm.download().reshape(C, T)[:64]
b = o_tm.download().reshape(T, C)[:, :64].T
print("equal:", np.array_equal(a, b, equal_nan=True))
