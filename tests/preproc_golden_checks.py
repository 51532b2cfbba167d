"""Checks shared by the CPU (oracle) and GPU (device) tests of the pre-processing steps against
tests/golden/preproc_deterministic.npz — the deterministic half of processing.jitter (processing.py:178-257) and
_processing._adapt_freq (_processing.py:75-142) as tests/golden/make_golden.py round3 computes it from the reference's own
kernels.  `impl` is `oracle` or `xsdba_amd.hip`: same function names and argument order."""
from __future__ import annotations

import numpy as np

from conftest import assert_close


def check_adapt_freq(impl, d, refuses_multiplicity=False):
    """refuses_multiplicity: the device entry point raises for a coarse group with a window (a time step twice in one block: an
    option combination the library says it does not compute) instead of computing it — those cases must raise."""
    n = int(d["n_af"])
    assert n == 6
    done = 0
    for i in range(n):
        k = f"af{i}"
        ref, sim, gidx = d[f"{k}_ref"], d[f"{k}_sim"], d[f"{k}_gidx"]
        (G, W), thresh = (int(v) for v in d[f"{k}_GW"]), float(d[f"{k}_thresh"])
        if refuses_multiplicity and G > 1 and G < 300 and W > 1:
            try:
                impl.adapt_freq(sim, ref, gidx, G, W, thresh, seed=1234 + i)
            except Exception as e:  # noqa: BLE001
                assert "occurs twice" in str(e)
                continue
            raise AssertionError("the device path was expected to refuse a coarse group with a window")
        out = impl.adapt_freq(sim, ref, gidx, G, W, thresh, seed=1234 + i)
        done += 1
        sim_ad, pth, dP0, P0_ref, P0_hist = (np.asarray(a) for a in out)
        # the proportions are ratios of two integer counts in float64: exact
        assert_close(P0_ref, d[f"{k}_P0_ref"], rtol=0.0, what=f"{k} P0_ref")
        assert_close(P0_hist, d[f"{k}_P0_hist"], rtol=0.0, what=f"{k} P0_hist")
        assert_close(dP0, d[f"{k}_dP0"], rtol=1e-15, what=f"{k} dP0")
        # np.nanquantile in float64, cast to the data dtype (nbutils.py:151-161)
        assert_close(pth, d[f"{k}_pth"], rtol=1e-6 if sim.dtype == np.float32 else 1e-13, what=f"{k} pth")
        # what the random half must respect: NaNs stay, values above the threshold are never touched, a replaced value lies
        # in [thresh, pth] of its group, groups that need no adaptation (dP0 <= 0 or NaN) are returned as they are
        assert np.array_equal(np.isnan(sim_ad), np.isnan(sim))
        changed = ~np.isnan(sim) & (sim_ad != sim)
        assert not (changed & (sim > thresh)).any(), k
        gdP0 = d[f"{k}_dP0"][:, gidx] if G > 1 else d[f"{k}_dP0"][:, np.zeros(sim.shape[1], int)]
        gpth = d[f"{k}_pth"][:, gidx] if G > 1 else d[f"{k}_pth"][:, np.zeros(sim.shape[1], int)]
        assert not (changed & ~(gdP0 > 0)).any(), k
        assert ((sim_ad[changed] >= thresh) & (sim_ad[changed] <= gpth[changed].astype(np.float64) * (1 + 1e-6))).all(), k
        if W == 1:
            # the NUMBER of replaced values per (cell, group) is fixed by the counts (the percent ranks are {(k-1)/(N-1)}): it is
            # only which of the tied values are drawn that is random.  (A draw can hit the old value itself only for thresh == value.)
            for g in range(G):
                sel = (gidx == g) if G > 1 else np.ones(sim.shape[1], bool)
                got = changed[:, sel].sum(axis=1)
                exp = d[f"{k}_nrep"][:, g]
                assert np.array_equal(got, exp), (k, g, got.tolist(), exp.tolist())
    assert done >= 4


def check_jitter(impl, d):
    n = int(d["n_jit"])
    assert n == 6
    for j in range(n):
        k = f"jit{j}"
        x = d[f"{k}_x"]
        lower, minimum, upper, maximum = (None if np.isnan(v) else float(v) for v in d[f"{k}_args"])
        out = np.asarray(impl.jitter(x, lower=lower, minimum=minimum, upper=upper, maximum=maximum, seed=99 + j))
        changed = ~np.isnan(x) & (out != x)
        exp_mask = np.zeros(x.shape, bool)
        if lower is not None:
            m = d[f"{k}_mask_lower"]
            exp_mask |= m
            jmin = float(d[f"{k}_jitter_min"])
            assert (out[m] >= jmin).all() and (out[m] <= lower).all(), k  # (U(jitter_min, lower) cast to the data dtype)
        if upper is not None:
            m = d[f"{k}_mask_upper"]
            exp_mask |= m
            jmax = float(d[f"{k}_jitter_max"])
            assert (out[m] >= upper).all() and (out[m] <= jmax).all(), k
        assert np.array_equal(np.isnan(out), np.isnan(x))
        # every value the reference replaces is replaced (a draw equal to the old value has probability ~0), nothing else is
        assert not (changed & ~exp_mask).any(), k
        assert (changed | ~exp_mask).mean() > 0.999, k
