"""Series of 22 529 .. 65 535 time steps (train_xlong.hpp: K time chunks sorted in LDS, merged as runs of (key, time) pairs in
global scratch, the walk on 14 / 9 waves with rank[] behind the L2) — the reference's nbutils.py:113-148 has no length limit.
EQM tables and plain group quantiles bit for bit against the oracle, DQM within its summation bound: two and three chunks, both
vector geometries (<= 45 056 steps / <= 65 536), the padded [G][256] toggle table (day-of-year windows on daily and hourly
steps) and the CSR lists (180 years of daily steps: 360 toggles per group; un-windowed months; 40 week groups), odd lengths,
ties across chunks, NaNs, a chunk without a valid value, an all-NaN cell, the key ranges of test_long_series_key_ranges."""
from __future__ import annotations

import numpy as np
import pytest

import oracle
from conftest import assert_close

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from xsdba_amd import hip as h

    h.default_context(0)
    return h


def _nodes(n):
    return ((np.arange(n) + 0.5) / n).astype(np.float32)


def _month_of(doy):
    cum = np.concatenate([[0], np.cumsum([31, 28, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31])])
    return (np.searchsorted(cum, doy, side="right") - 1).astype(np.int32)


GROUPINGS = {
    "doy31": lambda T: ((np.arange(T) % 365).astype(np.int32), 365, 31),
    "hourly_doy31": lambda T: (((np.arange(T) // 24) % 365).astype(np.int32), 365, 31),
    "month": lambda T: (_month_of(np.arange(T) % 365), 12, 1),
    "week40": lambda T: (((np.arange(T) // 7) % 40).astype(np.int32), 40, 1),
}


def _synth(rng, C, T, kind):
    doy = np.arange(T) % 365
    if kind == "+":
        off = rng.standard_normal((C, 1)) * 3
        a = 280 - 20 * np.cos(2 * np.pi * (doy + 1) / 365) + 2 * rng.random((C, T)) + 0.1 * np.arange(T) / 365 + off
        b = 280 - 18 * np.cos(2 * np.pi * (doy + 1) / 365) + 2 * rng.random((C, T)) + 0.11 * np.arange(T) / 365 + off
    else:
        a, b = rng.gamma(4, 7.5, (C, T)), rng.gamma(8.15, 3.68, (C, T))
    a, b = a.astype(np.float32), b.astype(np.float32)
    for x in (a, b):
        x[rng.random(x.shape) < 0.01] = np.nan
    return a, b


CASES = [(22529, "doy31"), (32768, "doy31"), (43800, "doy31"), (45056, "doy31"), (45057, "doy31"), (65535, "doy31"),
         (43800, "hourly_doy31"), (30001, "month"), (40000, "week40")]


@pytest.mark.parametrize("T,gname", CASES)
def test_xlong_series_train(hip, T, gname):
    rng = np.random.default_rng(T + len(gname))
    C = 3
    gidx, G, W = GROUPINGS[gname](T)
    q = _nodes(20)
    for kind in "+*":
        ref, hist = _synth(rng, C, T, kind)
        hist[0] = np.round(hist[0])          # ties, many of them across the chunks
        ref[2, : T // 2 + 7] = np.nan        # the first chunk (and a bit) missing
        hist[2, T // 2 - 7:] = np.nan        # the last chunk(s) missing
        if kind == "*":
            hist[1] = np.nan                 # an all-NaN cell
        af, hq = hip.eqm_train(ref, hist, gidx, G, W, q, kind)
        eaf, ehq = oracle.eqm_train(ref, hist, gidx, G, W, q, kind)
        assert_close(hq, ehq, rtol=0.0, what=f"hist_q {kind}")
        assert_close(af, eaf, rtol=0.0, what=f"af {kind}")
    got = hip.quantile_groups(hist, gidx, G, W, q)
    assert_close(got, oracle.quantile_groups(hist, gidx, G, W, q), rtol=0.0, what="quantile_groups")
    ref, hist = _synth(rng, C, T, "+")
    af, hq, sc = hip.dqm_train(ref, hist, gidx, G, W, q, "+")
    eaf, ehq, esc = oracle.dqm_train(ref, hist, gidx, G, W, q, "+")
    assert_close(sc, esc, rtol=2.4e-7, atol=2e-5, what="scaling")
    assert_close(hq, ehq, rtol=2.4e-7, atol=1e-4, what="hist_q (dqm)")
    assert_close(af, eaf, rtol=2.4e-7, atol=1e-4, what="af (dqm)")


@pytest.mark.parametrize("T", [43800, 60001])
def test_xlong_series_key_ranges(hip, T):
    """a series that spans zero (32-bit keys), precipitation with dry days (half the series one value: ties across every
    chunk), seven distinct values, a constant series, +-inf, every exponent, negative only; odd length"""
    rng = np.random.default_rng(T + 5)
    gidx, G, W = GROUPINGS["doy31"](T)
    q = _nodes(20)
    doy = np.arange(T) % 365
    C = 7
    x = np.empty((2, C, T), np.float32)
    for a in range(2):
        x[a, 0] = -2 + 15 * np.cos(2 * np.pi * doy / 365) + 4 * rng.standard_normal(T)
        x[a, 1] = rng.gamma(0.6, 5.0, T) * (rng.random(T) < 0.6)
        x[a, 2] = rng.integers(0, 7, T)
        x[a, 3] = 3.25
        x[a, 4] = 280 + 10 * rng.standard_normal(T)
        x[a, 4, rng.integers(0, T, 5)] = np.inf
        x[a, 4, rng.integers(0, T, 5)] = -np.inf
        x[a, 5] = np.sign(rng.standard_normal(T)) * 10.0 ** rng.uniform(-30, 30, T)
        x[a, 6] = -np.abs(100 + 30 * rng.standard_normal(T))
    x[:, :, rng.integers(0, T, 40)] = np.nan
    ref, hist = x[0], x[1]
    af, hq = hip.eqm_train(ref, hist, gidx, G, W, q, "+")
    eaf, ehq = oracle.eqm_train(ref, hist, gidx, G, W, q, "+")
    assert_close(hq, ehq, rtol=0.0, what="hist_q")
    assert_close(af, eaf, rtol=0.0, what="af")


def test_xlong_matches_generic_kernel(hip, monkeypatch):
    """the same tables as k_train_rows (XSDBA_HIP_TRAIN=rows), more cells than blocks resident on the device"""
    rng = np.random.default_rng(9)
    T, C = 30000, 300
    gidx, G, W = GROUPINGS["doy31"](T)
    q = _nodes(20)
    ref, hist = _synth(rng, C, T, "+")
    af, hq = hip.eqm_train(ref, hist, gidx, G, W, q, "+")
    monkeypatch.setenv("XSDBA_HIP_TRAIN", "rows")
    af2, hq2 = hip.eqm_train(ref, hist, gidx, G, W, q, "+")
    assert_close(hq, hq2, rtol=0.0, what="hist_q")
    assert_close(af, af2, rtol=0.0, what="af")


@pytest.mark.parametrize("T,gname", [(22529, "doy31"), (26000, "month"), (30001, "week40")])
def test_xlong_series_float64(hip, T, gname):
    """float64 series of these lengths take the generic row kernel (k_train_long / k_train_xlong are float32 kernels): a copy of
    the float32 dispatch once handed them doubles — EQM / DQM tables and group quantiles against the oracle"""
    rng = np.random.default_rng(T)
    C = 2
    gidx, G, W = GROUPINGS[gname](T)
    q = ((np.arange(20) + 0.5) / 20).astype(np.float64)
    ref, hist = (x.astype(np.float64) * (1 + 1e-9 * rng.random(x.shape)) for x in _synth(rng, C, T, "+"))
    af, hq = hip.eqm_train(ref, hist, gidx, G, W, q, "+")
    eaf, ehq = oracle.eqm_train(ref, hist, gidx, G, W, q, "+")
    assert np.asarray(hq).dtype == np.float64
    assert_close(hq, ehq, rtol=0.0, what="hist_q")
    assert_close(af, eaf, rtol=0.0, what="af")
    got = hip.quantile_groups(hist, gidx, G, W, q)
    assert_close(got, oracle.quantile_groups(hist, gidx, G, W, q), rtol=0.0, what="quantile_groups")
    af, hq, sc = hip.dqm_train(ref, hist, gidx, G, W, q, "+")
    eaf, ehq, esc = oracle.dqm_train(ref, hist, gidx, G, W, q, "+")
    assert_close(sc, esc, rtol=1e-12, atol=1e-10, what="scaling")   # (float64 block-reduction order of the group means)
    assert_close(hq, ehq, rtol=1e-12, atol=1e-10, what="hist_q (dqm)")
