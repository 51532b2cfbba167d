"""No length cliffs (the reference's nbutils._nan_quantile_1d sorts whatever it is given, nbutils.py:137; utils.rank and
PolyDetrend work on any series, utils.py:573-638, detrending.py:165-208):
  * group rows beyond the LDS sort — an hourly 30-year series under group="time" is ONE row of 262 800 values — go through the
    global-memory row pipeline (csrc/train_bigrows.hpp) instead of XSDBA_E_ROW_TOO_LONG: plain quantiles, EQM, DQM, vecquantiles,
    float32 and float64;
  * DQM and QDM train + adjust on series that do not fit the LDS (5 hourly years: 43 800 steps; 175 KB of float32): the trend and
    rank kernels read the series from global memory (k_group_trend<XG>, k_qdm_adjust<XG>).
Against the oracle: quantiles, EQM tables and ranks bit for bit, DQM within its summation bound."""
from __future__ import annotations

import numpy as np
import pytest

import oracle
from conftest import assert_close, assert_dqm_close_or_flip

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from xsdba_amd import hip as h

    h.default_context(0)
    return h


def _nodes(n, dtype=np.float32):
    return ((np.arange(n) + 0.5) / n).astype(dtype)


@pytest.mark.parametrize("T,dtype", [(262800, np.float32), (40000, np.float32), (65537, np.float32), (40000, np.float64), (20001, np.float64)])
def test_eqm_group_time_on_rows_beyond_the_lds(hip, T, dtype):
    rng = np.random.default_rng(T)
    C = 2
    ref = (280 + 9 * rng.standard_normal((C, T))).astype(dtype)
    hist = (281 + 8 * rng.standard_normal((C, T))).astype(dtype)
    hist[1, rng.random(T) < 0.03] = np.nan
    ref[0, :7] = ref[0, 7]                       # ties
    g = np.zeros(T, np.int32)
    q = _nodes(20, dtype)
    af, hq = hip.eqm_train(ref, hist, g, 1, 1, q, "+")
    eaf, ehq = oracle.eqm_train(ref, hist, g, 1, 1, q, "+")
    assert_close(hq, ehq, rtol=0.0, what="hist_q")
    assert_close(af, eaf, rtol=0.0, what="af")
    got = hip.quantile_groups(hist, g, 1, 1, q)
    assert_close(got, oracle.quantile_groups(hist, g, 1, 1, q), rtol=0.0, what="quantile_groups")
    if dtype == np.float32:
        np.testing.assert_allclose(got[0, 0], np.nanquantile(hist[0].astype(np.float64), q.astype(np.float64)), rtol=2e-6)
    sim = (282 + 8 * rng.standard_normal((C, T))).astype(dtype)
    scen = hip.qm_adjust(sim, af, hq, g, "nearest", "constant", "+")
    np.testing.assert_array_equal(scen, oracle.qm_adjust(sim, eaf, ehq, g, "nearest", "constant", "+"))


def test_long_rows_other_modes(hip):
    rng = np.random.default_rng(5)
    T, C = 70001, 3
    x = rng.gamma(2.0, 3.0, (C, T)).astype(np.float32)
    y = rng.gamma(2.5, 2.5, (C, T)).astype(np.float32)
    x[2] = np.nan                                 # an all-NaN row
    y[0, rng.random(T) < 0.1] = np.nan
    # two groups of unequal length, one of them beyond the LDS, and steps in no group
    g = (np.arange(T) % 5 == 0).astype(np.int32)
    g[::1001] = -1
    q = _nodes(20)
    af, hq, sc = hip.dqm_train(x, y, g, 2, 1, q, "*")
    eaf, ehq, esc = oracle.dqm_train(x, y, g, 2, 1, q, "*")
    assert_close(sc, esc, rtol=3e-7, what="scaling")
    assert_close(hq, ehq, rtol=3e-7, what="dqm hist_q")
    assert_close(af, eaf, rtol=6e-7, what="dqm af")
    rnk = np.array([0.0, 0.37, 0.99], np.float32)
    assert_close(hip.vecquantiles(y, rnk), oracle.vecquantiles(y, rnk) if hasattr(oracle, "vecquantiles") else
                 np.array([np.nanquantile(y[i].astype(np.float64), float(rnk[i])) for i in range(C)], np.float32), rtol=3e-7, what="vecquantiles")


def _hourly_doy(T):
    return ((np.arange(T) // 24) % 365).astype(np.int32)


@pytest.mark.parametrize("kind", ["+", "*"])
def test_dqm_on_a_series_beyond_the_lds(hip, kind):
    rng = np.random.default_rng(11)
    T, C, W = 43800, 3, 31
    g = _hourly_doy(T)
    day = np.arange(T) / 24.0
    if kind == "+":
        base = 280 - 15 * np.cos(2 * np.pi * day / 365)
        ref, hist, sim = (base + s * rng.standard_normal((C, T)) + d * day / 365 for s, d in ((2.0, 0.1), (2.2, 0.12), (2.1, 0.3)))
    else:
        ref, hist, sim = rng.gamma(4, 7.5, (C, T)), rng.gamma(8.15, 3.68, (C, T)), rng.gamma(16, 2.63, (C, T)) * (1 + 0.05 * day / 365)
    ref, hist, sim = (a.astype(np.float32) for a in (ref, hist, sim))
    sim[1, rng.random(T) < 0.02] = np.nan
    q = _nodes(20)
    tnum = (np.arange(T, dtype=np.float64) * 3600e9) + 946684800e9
    af, hq, sc = hip.dqm_train(ref, hist, g, 365, W, q, kind)
    eaf, ehq, esc = oracle.dqm_train(ref, hist, g, 365, W, q, kind)
    scale = float(np.nanmax(np.abs(ref))) if kind == "+" else 1.0
    assert_close(sc, esc, rtol=3e-7, atol=3e-7 * scale, what="scaling")
    assert_close(hq, ehq, rtol=3e-7, atol=3e-7 * scale, what="hist_q")
    for want_trend in (True, False):
        scen, trend = hip.dqm_adjust(sim, eaf, ehq, esc, g, W, tnum, 1, "nearest", "constant", kind, want_trend=want_trend)
        escen, etrend = oracle.dqm_adjust(sim, eaf, ehq, esc, g, W, tnum, 1, "nearest", "constant", kind)
        if want_trend:
            assert_close(trend, etrend, rtol=3e-7, atol=3e-7 * scale, what="trend")
        assert_dqm_close_or_flip(scen, escen, sim, esc, etrend, ehq, eaf, g, kind, rtol=1e-5, what=f"scen (trend array: {want_trend})")
    # PolyDetrend of degree 2 and the rolling mean of the whole series on the same long series
    tr2 = hip.polydetrend_trend(sim, g, 365, W, tnum, degree=2, kind=kind)
    assert tr2.shape == sim.shape and np.isfinite(tr2[0]).all()
    rm = hip.rolling_mean(sim[:1], 25)
    ker = np.ones(25) / 25
    np.testing.assert_allclose(rm[0, 12:-12], np.convolve(sim[0].astype(np.float64), ker, mode="valid"), rtol=2e-6)


@pytest.mark.parametrize("rank_window", [False, True])
def test_qdm_on_a_series_beyond_the_lds(hip, rank_window):
    rng = np.random.default_rng(13)
    T, C, W = 43800, 2, 5 if rank_window else 31
    g = _hourly_doy(T)
    ref, hist, sim = (rng.gamma(k, th, (C, T)).astype(np.float32) for k, th in ((4, 7.5), (8.15, 3.68), (16, 2.63)))
    sim[0, rng.random(T) < 0.02] = np.nan
    sim[1, 100:110] = sim[1, 100]               # ties
    q = _nodes(20)
    eaf, _ = oracle.eqm_train(ref, hist, g, 365, W, q, "*")
    scen, simq = hip.qdm_adjust(sim, eaf, q, g, W, rank_window, "nearest", "constant", "*")
    escen, esimq = oracle.qdm_adjust(sim, eaf, q, g, W, rank_window, "nearest", "constant", "*")
    np.testing.assert_array_equal(simq, esimq)   # ranks: bit-exact
    np.testing.assert_array_equal(scen, escen)
    ranks = hip.rank_groups(sim, g, 365, W, rank_window)
    np.testing.assert_array_equal(ranks, esimq)


@pytest.mark.parametrize("T,dtype", [(21900, np.float64), (40000, np.float64), (70001, np.float32)])
def test_loci_rows_beyond_the_lds(T, dtype):
    """LOCI under group="time": the hist row (every step of the series) goes through the global-memory row sort once it is
    longer than the LDS takes (32 768 float32 / 16 384 float64 values) — thresholds and factors against the oracle
    (reference _adjustment.py:889-919)"""
    from xsdba_amd import hip

    rng = np.random.default_rng(T)
    C = 2
    ref = rng.gamma(2.0, 4.0, (C, T)).astype(dtype)
    hist = rng.gamma(2.5, 3.0, (C, T)).astype(dtype)
    hist[1, rng.random(T) < 0.02] = np.nan
    ref[1, rng.random(T) < 0.02] = np.nan
    gidx = np.zeros(T, np.int32)
    af, th = hip.loci_train(ref, hist, gidx, 1, 1, 3.0)
    eaf, eth = oracle.loci_train(ref, hist, gidx, 1, 1, 3.0)
    np.testing.assert_array_equal(np.asarray(th), eth)
    np.testing.assert_allclose(np.asarray(af), eaf, rtol=1e-6 if dtype == np.float32 else 1e-12)
    # two groups of which one is long
    g2 = (np.arange(T) % 5 == 0).astype(np.int32)
    af, th = hip.loci_train(ref, hist, g2, 2, 1, 3.0)
    eaf, eth = oracle.loci_train(ref, hist, g2, 2, 1, 3.0)
    np.testing.assert_array_equal(np.asarray(th), eth)
    np.testing.assert_allclose(np.asarray(af), eaf, rtol=1e-6 if dtype == np.float32 else 1e-12)
