"""Large-block sanity check: 200 000 cells x 10950 days on one GPU, first / middle / last cells verified against the oracle
(guards against 32-bit index overflow in the kernels)."""
import sys, numpy as np
sys.path.insert(0, ".")
import oracle
from xsdba_amd import hip
C, T = int(sys.argv[1]) if len(sys.argv) > 1 else 200000, 10950
G, W, Q = 365, 31, 20
ctx = hip.default_context(0)
doy = (np.arange(T) % 365).astype(np.int32)
q = ((np.arange(Q) + 0.5) / Q).astype(np.float32)
tnum = (np.arange(T, dtype=np.float64) + 10950.0) * 86400e9
ref, hist, sim = (ctx.synth((C, T), np.float32, 0, i, 1, 0) for i in range(3))
af, hq, sc = hip.dqm_train(ref, hist, doy, G, W, q, "+", ctx=ctx)
scen, trend = hip.dqm_adjust(sim, af, hq, sc, doy, W, tnum, 1, "nearest", "constant", "+", ctx=ctx)
af2, hq2 = hip.eqm_train(ref, hist, doy, G, W, q, "+", ctx=ctx)
scen2, simq = hip.qdm_adjust(sim, af2, q, doy, W, True, "nearest", "constant", "+", ctx=ctx)
# round-2 paths: k_qdm_lanes (no rank_window, no sim_q) and the DQM mapping without a trend array
scen3, _ = hip.qdm_adjust(sim, af2, q, doy, W, False, "nearest", "constant", "+", want_sim_q=False, ctx=ctx)
scen4, _ = hip.dqm_adjust(sim, af, hq, sc, doy, W, tnum, 1, "nearest", "constant", "+", ctx=ctx, want_trend=False)
ctx.synchronize()
pick = [0, 1, C // 2, C - 2, C - 1]
def rows(d, shape_tail):
    full = d.download() if hasattr(d, "download") else d
    return np.stack([full[i] for i in pick])
R, H, S = rows(ref, None), rows(hist, None), rows(sim, None)
eaf, ehq, esc = oracle.dqm_train(R, H, doy, G, W, q, "+")
escen, etr = oracle.dqm_adjust(S, eaf, ehq, esc, doy, W, tnum, 1, "nearest", "constant", "+")
eaf2, ehq2 = oracle.eqm_train(R, H, doy, G, W, q, "+")
escen2, esq = oracle.qdm_adjust(S, eaf2, q, doy, W, True, "nearest", "constant", "+")
escen3, _ = oracle.qdm_adjust(S, eaf2, q, doy, W, False, "nearest", "constant", "+")
ok = True
for name, got, exp, tol in (("dqm af", rows(af, 0), eaf, 1e-5), ("dqm scen", rows(scen, 0), escen, 1e-5), ("eqm af", rows(af2, 0), eaf2, 0),
                            ("qdm scen", rows(scen2, 0), escen2, 0), ("sim_q", rows(simq, 0), esq, 0),
                            ("qdm scen (lanes)", rows(scen3, 0), escen3, 0), ("dqm scen (no trend array)", rows(scen4, 0), escen, 1e-5)):
    err = np.nanmax(np.abs(got.astype(np.float64) - exp) / np.maximum(1e-30, np.abs(exp)))
    print(name, "max rel err", err)
    ok &= bool(err <= tol) and np.array_equal(np.isnan(got), np.isnan(exp))
# round 4: the (time, cell) adjustment in place (k_qm_adjust_tm) on a cube of > 2^31 elements: selected cells against the oracle
for d in (scen, trend, scen2, simq, scen3, scen4, af, hq, sc):
    if hasattr(d, "free"):
        d.free()
sim_tm = ctx.synth((C, T), np.float32, 0, 2, 1, 0, time_major=True)
scen_tm = hip.qm_adjust(sim_tm, af2, hq2, doy, "nearest", "constant", "+", ctx=ctx, out_time_major=True)
ctx.synchronize()
from xsdba_amd._lib import DeviceArray
cols = np.stack([np.concatenate([DeviceArray.from_ptr(ctx, scen_tm.ptr + (t * C + c) * 4, (1,), np.float32).download() for t in range(0, T, 97)]) for c in pick])
escen_tm = oracle.qm_adjust(S, eaf2, ehq2, doy, "nearest", "constant", "+")[:, ::97]
print("eqm scen (time-major, every 97th step)", "equal" if np.array_equal(cols, escen_tm, equal_nan=True) else "MISMATCH")
ok &= np.array_equal(cols, escen_tm, equal_nan=True)
print("OK" if ok else "MISMATCH")
