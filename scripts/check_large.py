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
print("OK" if ok else "MISMATCH")
