"""GPU parity of the histogram multi-select training kernel (k_train_sel, xsdba_amd/csrc/train_select.hpp) that serves
un-windowed groups (group="time" / "time.month" / "time.season", BASELINE configs[0]/[1]).

It replaces nbutils._nan_quantile_1d's sort (nbutils.py:113-148) by counting + selection, so the bar is bit-exactness
against the oracle on every distribution that stresses the selection: zero-inflated and capped data (runs of the
extreme values), rounded data (runs of equal values inside bins), tiny integer ranges (one key per bin), huge dynamic
ranges (many distinct keys in one bin -> the kernel hands the cell to the generic kernel), NaNs, infinities, time steps
without a group, odd lengths (unaligned rows), every register-tile size.  Every case is also run with the sort-based
kernels pinned (XSDBA_HIP_TRAIN=sort) so that the three implementations are held to the same tables.
"""
from __future__ import annotations

import zlib

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


@pytest.fixture(params=["select", "sort"])
def kernel(request, monkeypatch):
    if request.param == "sort":
        monkeypatch.setenv("XSDBA_HIP_TRAIN", "sort")
    else:
        monkeypatch.delenv("XSDBA_HIP_TRAIN", raising=False)
    return request.param


def _month(T, start=0):
    d0 = (np.arange(T) + start) % 365
    cum = np.concatenate([[0], np.cumsum([31, 28, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31])])
    return (np.searchsorted(cum, d0, side="right") - 1).astype(np.int32)


def _season(T):
    return ((_month(T) + 1) % 12 // 3).astype(np.int32)


def _groups(name, T):
    if name == "time":
        return np.zeros(T, np.int32), 1
    if name == "month":
        return _month(T), 12
    if name == "season":
        return _season(T), 4
    if name == "g32":
        return (np.arange(T) % 32).astype(np.int32), 32
    raise KeyError(name)


def _nodes(n):
    return ((np.arange(n) + 0.5) / n).astype(np.float32)


def _check(hip, ref, hist, gidx, G, q, kind):
    af, hq = hip.eqm_train(ref, hist, gidx, G, 1, q, kind)
    eaf, ehq = oracle.eqm_train(ref, hist, gidx, G, 1, q, kind)
    assert_close(hq, ehq, rtol=0.0, what="hist_q")
    assert_close(af, eaf, rtol=0.0, what="af")
    got = hip.quantile_groups(ref, gidx, G, 1, q)
    exp = oracle.quantile_groups(ref, gidx, G, 1, q)
    assert_close(got, exp, rtol=0.0, what="quantile_groups")


DISTS = ["normal", "gamma", "zero_inflated", "capped", "rounded", "small_int", "constant", "huge_range", "inf", "tiny_n",
         "two_values", "denormal"]


def _draw(rng, dist, C, T):
    def one():
        if dist == "normal":
            a = 280 + 10 * rng.standard_normal((C, T))
        elif dist == "gamma":
            a = rng.gamma(0.7, 6.0, (C, T))
        elif dist == "zero_inflated":  # precipitation: half the days exactly dry
            a = rng.gamma(0.7, 6.0, (C, T)) * (rng.random((C, T)) < 0.5)
        elif dist == "capped":  # relative humidity: a mass at the maximum
            a = np.minimum(100.0, 80 + 15 * rng.standard_normal((C, T)))
        elif dist == "rounded":  # 0.1 resolution: runs of equal values everywhere
            a = np.round(rng.gamma(0.7, 6.0, (C, T)) * (rng.random((C, T)) < 0.7), 1)
        elif dist == "small_int":  # a few dozen distinct keys
            a = rng.integers(0, 37, (C, T)).astype(np.float64) * 2.0 ** -140
        elif dist == "constant":
            a = np.full((C, T), 3.25)
        elif dist == "huge_range":  # thousands of neighbouring floats in one bin + far outliers: the generic kernel redoes it
            a = 1.0 + 1.1920929e-07 * rng.integers(0, 3000, (C, T))
            a[:, ::997] = 1e30
            a[:, 5::1013] = -1e30
        elif dist == "inf":
            a = rng.standard_normal((C, T))
            a[rng.random((C, T)) < 0.01] = np.inf
            a[rng.random((C, T)) < 0.01] = -np.inf
        elif dist == "tiny_n":  # almost everything missing: groups of 0, 1, 2 ... valid values
            a = rng.standard_normal((C, T))
            a[rng.random((C, T)) < 0.995] = np.nan
        elif dist == "two_values":
            a = np.where(rng.random((C, T)) < 0.3, -1.5, 2.5)
        elif dist == "denormal":
            a = rng.standard_normal((C, T)) * 1e-42
        else:
            raise KeyError(dist)
        return a.astype(np.float32)

    return one(), one()


@pytest.mark.parametrize("gname", ["time", "month", "season"])
@pytest.mark.parametrize("dist", DISTS)
def test_select_bit_exact_vs_oracle(hip, kernel, gname, dist):
    rng = np.random.default_rng(zlib.crc32(f"{gname}{dist}".encode()))
    C, T = 6, 3650
    gidx, G = _groups(gname, T)
    ref, hist = _draw(rng, dist, C, T)
    if dist not in ("tiny_n",):
        ref[rng.random(ref.shape) < 0.02] = np.nan
        hist[1, : T // 2] = np.nan
        ref[2] = np.nan  # an all-NaN cell
    kind = "*" if dist in ("gamma", "zero_inflated", "rounded") else "+"
    _check(hip, ref, hist, gidx, G, _nodes(20), kind)


@pytest.mark.parametrize("T", [64, 65, 2047, 2049, 4097, 8193, 10950, 11263, 11265, 16385, 21900, 22528])
def test_select_lengths(hip, kernel, T):
    """every register tile (2 .. 22 float2 loads per thread), odd lengths (rows that are only 4-byte aligned)"""
    rng = np.random.default_rng(T)
    C = 3
    gidx, G = _groups("month", T)
    ref = (280 + 10 * rng.standard_normal((C, T))).astype(np.float32)
    hist = (281 + 9 * rng.standard_normal((C, T))).astype(np.float32)
    hist[rng.random(hist.shape) < 0.01] = np.nan
    _check(hip, ref, hist, gidx, G, _nodes(20), "+")


@pytest.mark.parametrize("Q", [1, 2, 15, 31, 50])
@pytest.mark.parametrize("gname", ["time", "month", "g32"])
def test_select_quantile_counts(hip, kernel, Q, gname):
    rng = np.random.default_rng(Q)
    C, T = 3, 5000
    gidx, G = _groups(gname, T)
    ref = rng.gamma(2.0, 3.0, (C, T)).astype(np.float32)
    hist = rng.gamma(2.5, 2.5, (C, T)).astype(np.float32)
    q = _nodes(Q) if Q > 2 else np.array([0.0, 1.0][:Q], np.float32)
    _check(hip, ref, hist, gidx, G, q, "*")


def test_select_unsorted_and_repeated_quantiles(hip, kernel):
    """ordinals that repeat or come unsorted share bins in any order"""
    rng = np.random.default_rng(5)
    C, T = 3, 4000
    gidx, G = _groups("month", T)
    ref = rng.standard_normal((C, T)).astype(np.float32)
    hist = rng.standard_normal((C, T)).astype(np.float32)
    q = np.array([0.9, 0.1, 0.5, 0.5, 0.5001, 0.0, 1.0, 0.1, 0.25], np.float32)
    _check(hip, ref, hist, gidx, G, q, "+")


def test_select_ungrouped_steps(hip, kernel):
    """time steps that belong to no group (group index -1) are left out of every row"""
    rng = np.random.default_rng(6)
    C, T = 4, 3000
    gidx, G = _groups("season", T)
    gidx = gidx.copy()
    gidx[rng.random(T) < 0.2] = -1
    ref = rng.standard_normal((C, T)).astype(np.float32)
    hist = rng.standard_normal((C, T)).astype(np.float32)
    _check(hip, ref, hist, gidx, G, _nodes(20), "+")


def test_select_many_cells_device_resident(hip):
    """more cells than resident blocks (grid-stride walk), device-resident inputs and outputs"""
    rng = np.random.default_rng(8)
    C, T = 1500, 1200
    gidx, G = _groups("month", T)
    ref = (rng.gamma(0.8, 5.0, (C, T)) * (rng.random((C, T)) < 0.6)).astype(np.float32)
    hist = (rng.gamma(0.9, 4.0, (C, T)) * (rng.random((C, T)) < 0.7)).astype(np.float32)
    q = _nodes(20)
    af, hq = hip.eqm_train(ref, hist, gidx, G, 1, q, "*")
    eaf, ehq = oracle.eqm_train(ref, hist, gidx, G, 1, q, "*")
    assert_close(hq, ehq, rtol=0.0, what="hist_q")
    assert_close(af, eaf, rtol=0.0, what="af")


@pytest.mark.parametrize("gname", ["time", "month", "season"])
@pytest.mark.parametrize("dist,kind", [("normal", "+"), ("gamma", "*"), ("zero_inflated", "*"), ("rounded", "*"), ("small_int", "+"),
                                       ("constant", "+"), ("inf", "+"), ("two_values", "*"), ("tiny_n", "+"), ("capped", "*")])
def test_select_dqm_vs_oracle(hip, kernel, gname, dist, kind):
    """dqm_train on un-windowed groups (_adjustment.py:150-190): group-mean normalisation + quantiles + scaling.  Within 2 ulp
    of the oracle (float64 sums in another order); cells the shortcut cannot serve (negative multiplicative means, infinite
    values) come back from the generic kernel."""
    rng = np.random.default_rng(zlib.crc32(f"dqm{gname}{dist}".encode()))
    C, T = 6, 3650
    gidx, G = _groups(gname, T)
    ref, hist = _draw(rng, dist, C, T)
    if dist != "tiny_n":
        ref[rng.random(ref.shape) < 0.02] = np.nan
        hist[1, : T // 2] = np.nan
        ref[2] = np.nan
    q = _nodes(20)
    af, hq, sc = hip.dqm_train(ref, hist, gidx, G, 1, q, kind)
    with np.errstate(all="ignore"):
        eaf, ehq, esc = oracle.dqm_train(ref, hist, gidx, G, 1, q, kind)
    scale = float(np.nanmax(np.abs(np.where(np.isfinite(ehq), ehq, np.nan)))) if np.isfinite(ehq).any() else 1.0
    assert_close(sc, esc, rtol=3e-7, atol=1e-30, what="scaling")
    assert_close(hq, ehq, rtol=3e-7, atol=3e-7 * scale, what="hist_q")
    escale = float(np.nanmax(np.abs(np.where(np.isfinite(eaf), eaf, np.nan)))) if np.isfinite(eaf).any() else 1.0
    assert_close(af, eaf, rtol=2e-6, atol=2e-6 * max(scale, escale), what="af")
