"""Random configurations of the hot path against the oracle (bit-exact EQM / QDM, DQM within 2e-6): series length, grouping,
window, quantile count, kind, NaNs, ties (rounded values, zero-inflation, constant stretches), rank_window, extrapolation.
    python scripts/diag/fuzz_vs_oracle.py [n_cases] [seed]
Prints the failing configurations (none expected) and a count."""
import sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import oracle
from xsdba_amd import hip

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
long_series = len(sys.argv) > 3 and sys.argv[3] == "long"   # also series beyond 22 528 steps (the chunk-merge kernel)
f64_share = 0.25
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)


def doy_month(T, start):
    d0 = (np.arange(T) + start) % 365
    cum = np.concatenate([[0], np.cumsum([31, 28, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31])])
    return d0.astype(np.int32), (np.searchsorted(cum, d0, side="right") - 1).astype(np.int32)


def series(C, T, kind, dtype):
    if kind == "+":
        x = 280 + 10 * rng.standard_normal((C, T))
    else:
        x = rng.gamma(2.0, 4.0, (C, T))
    mode = rng.integers(0, 6)
    if mode == 1:
        x = np.round(x, int(rng.integers(0, 3)))
    elif mode == 2 and kind == "*":
        x[rng.random((C, T)) < rng.uniform(0.2, 0.8)] = 0.0
    elif mode == 3:
        x[:, : T // 3] = x[:, :1]
    elif mode == 4:
        x = np.minimum(x, np.quantile(x, 0.7))
    x = x.astype(dtype)
    if rng.random() < 0.5:
        x[rng.random((C, T)) < rng.uniform(0.0, 0.1)] = np.nan
    if rng.random() < 0.2:
        x[0] = np.nan
    return x


bad = 0
for case in range(n_cases):
    T = int(rng.choice([400, 730, 1500, 3650, 4500, 7300, 9000, 10950, 12000, 16000, 21900] + ([26000] if long_series else [])))
    C = 3
    doy, mon = doy_month(T, int(rng.integers(0, 365)))
    gname = rng.choice(["time", "month", "season", "doy"])
    if gname == "time":
        gidx, G, W = np.zeros(T, np.int32), 1, 1
    elif gname == "month":
        gidx, G, W = mon, 12, int(rng.choice([1, 1, 3]))
    elif gname == "season":
        gidx, G, W = ((mon + 1) % 12 // 3).astype(np.int32), 4, 1
    else:
        gidx, G, W = doy, 365, int(rng.choice([1, 5, 15, 31]))
    if rng.random() < 0.15 and G > 1:
        gidx = gidx.copy(); gidx[rng.random(T) < 0.03] = -1
    Q = int(rng.choice([5, 15, 20, 20, 33, 50]))
    dtype = np.float64 if rng.random() < f64_share else np.float32
    q = ((np.arange(Q) + 0.5) / Q).astype(dtype)
    kind = str(rng.choice(["+", "*"]))
    extrap = str(rng.choice(["constant", "nan"]))
    ref, hist, sim = series(C, T, kind, dtype), series(C, T, kind, dtype), series(C, T, kind, dtype)
    layout = "cell-major"
    if rng.random() < 0.3:  # the same cubes stored (time, cell): the wrappers hand element strides over
        ref, hist, sim = (np.asfortranarray(x) for x in (ref, hist, sim))
        layout = "time-major"
    tag = f"case {case}: T={T} group={gname} W={W} Q={Q} kind={kind} extrap={extrap} {np.dtype(dtype).name} {layout}"
    try:
        af, hq = hip.eqm_train(ref, hist, gidx, G, W, q, kind)
        eaf, ehq = oracle.eqm_train(ref, hist, gidx, G, W, q, kind)
        np.testing.assert_array_equal(np.asarray(hq), ehq, err_msg="hist_q")
        np.testing.assert_array_equal(np.asarray(af), eaf, err_msg="af")
        scen = hip.qm_adjust(sim, eaf, ehq, gidx, "nearest", extrap, kind)
        escen = oracle.qm_adjust(sim, eaf, ehq, gidx, "nearest", extrap, kind)
        ok = np.isclose(np.asarray(scen), escen, rtol=1e-6, atol=0, equal_nan=True)
        if not ok.all():  # exact distance ties between rows are implementation-defined: count, do not fail, unless many
            assert (~ok).mean() < 1e-3, f"qm_adjust {(~ok).sum()} mismatches"
        for rw in ((False, True) if W > 1 else (False,)):
            s2, sq = hip.qdm_adjust(sim, eaf, q, gidx, W, rw, "nearest", extrap, kind)
            es2, esq = oracle.qdm_adjust(sim, eaf, q, gidx, W, rw, "nearest", extrap, kind)
            np.testing.assert_array_equal(np.asarray(sq), esq, err_msg=f"sim_q rw={rw}")
            np.testing.assert_array_equal(np.asarray(s2), es2, err_msg=f"qdm scen rw={rw}")
            s3, _ = hip.qdm_adjust(sim, eaf, q, gidx, W, rw, "nearest", extrap, kind, want_sim_q=False)
            np.testing.assert_array_equal(np.asarray(s3), es2, err_msg=f"qdm scen (no sim_q) rw={rw}")
        if kind == "+" or (np.nanmin(ref) > 0 and np.nanmin(hist) > 0):
            a2, h2, s2_ = hip.dqm_train(ref, hist, gidx, G, W, q, kind)
            ea2, eh2, es2_ = oracle.dqm_train(ref, hist, gidx, G, W, q, kind)
            np.testing.assert_allclose(np.asarray(s2_), es2_, rtol=2e-6, equal_nan=True, err_msg="dqm scaling")
            np.testing.assert_allclose(np.asarray(h2), eh2, rtol=4e-6, atol=1e-6, equal_nan=True, err_msg="dqm hist_q")
        # the adjust side of DQM (scaling, per-group polynomial trend, mapping, retrend), group means, LOCI, adapt_freq's tables
        tnum = (np.arange(T, dtype=np.float64) * 86400e9 + 9.4e17)
        if kind == "+" or (np.nanmin(sim) > 0 and np.nanmin(hist) > 0 and np.nanmin(ref) > 0):
            ea2, eh2, es2_ = oracle.dqm_train(ref, hist, gidx, G, W, q, kind)
            deg = int(rng.choice([0, 1, 1]))
            scn = hip.dqm_adjust(sim, ea2, eh2, es2_, gidx, W, tnum, deg, "nearest", extrap, kind)
            scn = scn[0] if isinstance(scn, tuple) else scn
            escn = oracle.dqm_adjust(sim, ea2, eh2, es2_, gidx, W, tnum, deg, "nearest", extrap, kind)
            escn = escn[0] if isinstance(escn, tuple) else escn
            okd = np.isclose(np.asarray(scn), escn, rtol=2e-5 if dtype == np.float32 else 1e-9, atol=1e-4 if dtype == np.float32 else 1e-9, equal_nan=True)
            assert (~okd).mean() < 2e-3, f"dqm_adjust {(~okd).sum()} of {okd.size} mismatches (degree {deg})"
        gm = hip.group_means(sim, gidx, G, W)
        np.testing.assert_allclose(np.asarray(gm), oracle.group_means(sim, gidx, G, W), rtol=1e-6 if dtype == np.float32 else 1e-12, equal_nan=True, err_msg="group_means")
        if kind == "*":
            th = float(np.nanquantile(hist, 0.3)) if np.isfinite(hist).any() else 1.0
            laf, lth = hip.loci_train(ref, hist, gidx, G, W, th)
            elaf, elth = oracle.loci_train(ref, hist, gidx, G, W, th)
            np.testing.assert_allclose(np.asarray(lth), elth, rtol=1e-6, equal_nan=True, err_msg="loci hist_thresh")
            # (hist values above the threshold all equal to it — capped data — make the factor a division by the rounding
            #  residue of a mean, 1e14 or inf as the summation order has it: not compared)
            sane = ~(np.abs(elaf) > 1e9) & ~(np.abs(np.asarray(laf)) > 1e9)
            np.testing.assert_allclose(np.asarray(laf)[sane], elaf[sane], rtol=2e-5, equal_nan=True, err_msg="loci af")
            if W == 1 or gname == "doy":  # (a coarse group with a window puts a step several times into one block: refused)
                import os
                sel = rng.random() < 0.6   # by selection over the row in global memory (the default), else the LDS sorts where they fit
                if not sel:
                    os.environ["XSDBA_HIP_AF_SELECT"] = "0"
                try:
                    out = hip.adapt_freq(sim, ref, gidx, G, W, th, seed=3)
                finally:
                    os.environ.pop("XSDBA_HIP_AF_SELECT", None)
                eout = oracle.adapt_freq(sim, ref, gidx, G, W, th, seed=3)
                np.testing.assert_array_equal(np.asarray(out[0]), eout[0], err_msg=f"adapt_freq sim_ad (select={sel})")
                for nm, a_, b_ in zip(("pth", "dP0", "P0_ref", "P0_hist"), out[1:], eout[1:]):
                    np.testing.assert_allclose(np.asarray(a_), b_, rtol=1e-6, equal_nan=True, err_msg=f"adapt_freq {nm}")
        if G == 1:
            for interp in ("linear",):
                s1 = hip.qm_adjust(sim, eaf, ehq, gidx, interp, extrap, kind)
                e1 = oracle.qm_adjust(sim, eaf, ehq, gidx, interp, extrap, kind)
                ok1 = np.isclose(np.asarray(s1), e1, rtol=1e-5, atol=1e-5, equal_nan=True)
                assert (~ok1).mean() < 1e-3, f"qm_adjust 1-D {interp}: {(~ok1).sum()} mismatches"
    except AssertionError as e:
        bad += 1
        print("FAIL", tag, "|", str(e).strip().splitlines()[:4])
        import os
        os.makedirs("gpurun_out", exist_ok=True)
        np.savez_compressed(f"gpurun_out/fuzz_fail_{case}.npz", ref=ref, hist=hist, sim=sim, gidx=gidx, G=G, W=W, q=q, kind=kind, extrap=extrap)
    except Exception as e:  # noqa: BLE001
        bad += 1
        print("ERROR", tag, "|", type(e).__name__, str(e)[:200])
    if case % 20 == 19:
        print(f"... {case + 1} cases, {bad} bad", flush=True)
print(f"done: {n_cases} cases, {bad} bad")
