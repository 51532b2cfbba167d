"""The per-cell sort behind the windowed training kernels (train_cell.hpp: one 14-bit counting pass on packed LDS bins, then
ceil(R / 8) stable counting passes on wave-private counter rows for the R key bits left — round 4; the last stable pass also
leaves rank[]): series whose keys span fewer than 14 bits (no stable pass at all: rank[] comes from the fallback loop), EVERY
width from 13 to 32 bits (one, two and three stable passes of every digit width, the 32-bit case of a series that spans zero),
one value only, one valid value only — EQM / DQM tables and QDM ranks bit for bit against the oracle, float32, the sizes of all
three EPT instantiations of the short kernel and the long-series kernel (keys in registers, 15-bit time steps, bit 31 beside
the handles)."""
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


def _doy(T):
    return (np.arange(T) % 365).astype(np.int32)


def _series(rng, kind, C, T):
    base = np.float32(280.0)
    ulp = np.spacing(base)
    if kind == "narrow":      # < 2^13 distinct keys: the counting pass alone sorts them
        return (base + ulp * rng.integers(0, 5000, (C, T))).astype(np.float32)
    if kind == "bits14":      # the range needs exactly 14 bits
        return (base + ulp * rng.integers(0, 1 << 14, (C, T))).astype(np.float32)
    if kind == "bits18":      # 14 + one nibble pass
        return (base + ulp * rng.integers(0, 1 << 18, (C, T))).astype(np.float32)
    if kind == "bits19":      # 14 + two nibble passes, the second over one bit
        return (base + ulp * rng.integers(0, 1 << 19, (C, T))).astype(np.float32)
    if kind == "wide":        # both signs, 60 binades: 32-bit keys
        return (rng.standard_normal((C, T)) * np.exp2(rng.integers(-30, 30, (C, T)))).astype(np.float32)
    if kind == "constant":
        return np.full((C, T), 3.25, np.float32)
    if kind == "one_valid":
        a = np.full((C, T), np.nan, np.float32)
        a[:, T // 3] = 1.5
        return a
    raise KeyError(kind)


KINDS = ["narrow", "bits14", "bits18", "bits19", "wide", "constant", "one_valid"]


@pytest.mark.parametrize("T", [3650, 7300, 10950])
@pytest.mark.parametrize("kind", KINDS)
def test_eqm_tables_on_edge_key_ranges(hip, kind, T):
    rng = np.random.default_rng(len(kind) * 1000 + T)
    C = 3
    ref, hist = _series(rng, kind, C, T), _series(rng, kind, C, T)
    if kind not in ("constant", "one_valid"):
        hist[1, rng.random(T) < 0.1] = np.nan
    q = ((np.arange(20) + 0.5) / 20).astype(np.float32)
    gidx = _doy(T)
    af, hq = hip.eqm_train(ref, hist, gidx, 365, 31, q, "+")
    eaf, ehq = oracle.eqm_train(ref, hist, gidx, 365, 31, q, "+")
    assert_close(hq, ehq, rtol=0.0, what="hist_q")
    assert_close(af, eaf, rtol=0.0, what="af")


@pytest.mark.parametrize("kind", ["narrow", "bits18", "wide", "constant"])
def test_qdm_ranks_on_edge_key_ranges(hip, kind):
    rng = np.random.default_rng(77 + len(kind))
    C, T = 3, 3650
    ref, hist, sim = (_series(rng, kind, C, T) for _ in range(3))
    q = ((np.arange(20) + 0.5) / 20).astype(np.float32)
    gidx = _doy(T)
    af, hq = hip.eqm_train(ref, hist, gidx, 365, 31, q, "+")
    for rw in (False, True):
        scen, sim_q = hip.qdm_adjust(sim, af, q, gidx, 31, rw, "nearest", "constant", "+")
        escen, esim_q = oracle.qdm_adjust(sim, af, q, gidx, 31, rw, "nearest", "constant", "+")
        assert_close(sim_q, esim_q, rtol=0.0, what=f"sim_q rank_window={rw}")
        assert_close(scen, escen, rtol=0.0, what=f"scen rank_window={rw}")


@pytest.mark.parametrize("kind", ["narrow", "bits19", "constant"])
def test_dqm_tables_on_edge_key_ranges(hip, kind):
    rng = np.random.default_rng(5 + len(kind))
    C, T = 3, 3650
    ref, hist = _series(rng, kind, C, T), _series(rng, kind, C, T)
    q = ((np.arange(20) + 0.5) / 20).astype(np.float32)
    gidx = _doy(T)
    af, hq, sc = hip.dqm_train(ref, hist, gidx, 365, 31, q, "+")
    eaf, ehq, esc = oracle.dqm_train(ref, hist, gidx, 365, 31, q, "+")
    atol = 2.4e-7 * 280
    assert_close(sc, esc, rtol=2.4e-7, atol=atol, what="scaling")
    assert_close(hq, ehq, rtol=2.4e-7, atol=atol, what="hist_q")
    assert_close(af, eaf, rtol=5e-7, atol=2 * atol, what="af")


def _width_series(rng, bits, C, T):
    """float32 values whose order-preserving keys span exactly `bits` bits: integers scaled by a power of two at a base whose ulp
    is known (<= 23 bits), else several binades (24 .. 31) and both signs (32)"""
    if bits <= 23:
        base = np.float32(512.0)  # ulp 2^-14 up to 1024: 2^23 steps stay inside the binade
        k = rng.integers(0, 1 << bits, (C, T))
        k[:, 0], k[:, 1] = 0, (1 << bits) - 1  # the full width is present in every series
        return (base + np.spacing(base) * k.astype(np.float64)).astype(np.float32)
    if bits <= 31:
        # positive floats over 2^(bits - 23) binades: keys of positive floats are their bit patterns | 0x80000000
        lo = np.float32(1.0).view(np.uint32).astype(np.int64)
        k = lo + rng.integers(0, (1 << bits) - 1, (C, T))
        k[:, 0], k[:, 1] = lo, lo + (1 << bits) - 2
        return k.astype(np.uint32).view(np.float32)
    x = (rng.standard_normal((C, T)) * np.exp2(rng.integers(-20, 20, (C, T)))).astype(np.float32)
    x[:, 0], x[:, 1] = -1e30, 1e30
    return x


@pytest.mark.parametrize("bits", list(range(13, 33)))
def test_every_key_width_short_and_long(hip, bits):
    """one 14-bit pass + 0 .. 3 stable passes of 1 .. 8 bits (short kernel, T = 4 000: EPT 4) and the long-series kernel
    (T = 12 000: 12 keys per thread in registers); ties across passes (5 % of the values repeated), NaNs"""
    rng = np.random.default_rng(bits)
    q = ((np.arange(20) + 0.5) / 20).astype(np.float32)
    for T in (4000, 12000):
        C = 2
        ref, hist = _width_series(rng, bits, C, T), _width_series(rng, bits, C, T)
        rep = rng.random(T) < 0.05
        hist[:, rep] = hist[:, 2:3]
        ref[1, rng.random(T) < 0.03] = np.nan
        gidx = _doy(T)
        af, hq = hip.eqm_train(ref, hist, gidx, 365, 31, q, "+")
        eaf, ehq = oracle.eqm_train(ref, hist, gidx, 365, 31, q, "+")
        assert_close(hq, ehq, rtol=0.0, what=f"hist_q bits={bits} T={T}")
        assert_close(af, eaf, rtol=0.0, what=f"af bits={bits} T={T}")


def test_partial_year_series_empty_middle_groups(hip):
    """Day-of-year groups on series that populate part of the year only (and group indexes with -1 gaps): a wave of the
    three-stream kernel whose MIDDLE group has an empty row still has to apply the toggle lists of the populated groups beside
    it (train_cell3.hpp: the early list requests sit inside the middle row's loop).  The end of the data is swept over every
    offset of a wave's 23 groups, from both sides; EQM, DQM and plain quantiles against the oracle, bit for bit."""
    rng = np.random.default_rng(11)
    q = ((np.arange(20) + 0.5) / 20).astype(np.float32)
    C = 2
    for start, T in [(0, t) for t in range(188, 236, 3)] + [(170, t) for t in range(100, 148, 5)]:
        gidx = ((np.arange(T) + start) % 365).astype(np.int32)
        ref = (280 + 5 * rng.standard_normal((C, T))).astype(np.float32)
        hist = (281 + 4 * rng.standard_normal((C, T))).astype(np.float32)
        af, hq = hip.eqm_train(ref, hist, gidx, 365, 31, q, "+")
        eaf, ehq = oracle.eqm_train(ref, hist, gidx, 365, 31, q, "+")
        assert_close(hq, ehq, rtol=0.0, what=f"hist_q start={start} T={T}")
        assert_close(af, eaf, rtol=0.0, what=f"af start={start} T={T}")
    # gaps in the group index: a block of days belongs to no group, its neighbours' windows still reach into it
    T = 1460
    gidx = (np.arange(T) % 365).astype(np.int32)
    gidx[(gidx >= 200) & (gidx < 231)] = -1
    ref = (280 + 5 * rng.standard_normal((C, T))).astype(np.float32)
    hist = (281 + 4 * rng.standard_normal((C, T))).astype(np.float32)
    af, hq = hip.eqm_train(ref, hist, gidx, 365, 31, q, "+")
    eaf, ehq = oracle.eqm_train(ref, hist, gidx, 365, 31, q, "+")
    assert_close(hq, ehq, rtol=0.0, what="hist_q with gaps")
    assert_close(af, eaf, rtol=0.0, what="af with gaps")
    af, hq, sc = hip.dqm_train(ref, hist, gidx, 365, 31, q, "+")
    eaf, ehq, esc = oracle.dqm_train(ref, hist, gidx, 365, 31, q, "+")
    assert_close(sc, esc, rtol=2.4e-7, atol=1e-4, what="scaling with gaps")
    assert_close(hq, ehq, rtol=2.4e-7, atol=1e-4, what="dqm hist_q with gaps")


@pytest.mark.parametrize("T", [3650, 7300, 10950])
@pytest.mark.parametrize("kind", ["zero_half", "two_values", "zero_most_wide", "runs_of_64"])
def test_heavily_tied_series_with_wide_keys(hip, kind, T):
    """hundreds of equal keys in a series whose key range still needs the stable passes: whole waves of one digit take their 64
    slots with one atomic (the sort's `heavy` path) — EQM / DQM tables and QDM ranks bit for bit against the oracle"""
    rng = np.random.default_rng(len(kind) * 77 + T)
    C = 3

    def make():
        a = rng.gamma(2.0, 8.0, (C, T)).astype(np.float32) + np.float32(1e-3)
        if kind == "zero_half":
            a[rng.random((C, T)) < 0.5] = 0.0
        elif kind == "two_values":
            a[rng.random((C, T)) < 0.45] = 0.0
            a[rng.random((C, T)) < 0.45] = 7.5
        elif kind == "zero_most_wide":   # 90 % zeros, the rest over 60 binades of both signs
            w = (rng.standard_normal((C, T)) * np.exp2(rng.integers(-30, 30, (C, T)))).astype(np.float32)
            a = np.where(rng.random((C, T)) < 0.9, np.float32(0.0), w)
        elif kind == "runs_of_64":       # every value 64 .. 200 times
            vals = rng.gamma(2.0, 8.0, 200).astype(np.float32)
            a = vals[rng.integers(0, max(T // 100, 2), (C, T)) % 200]
        return np.ascontiguousarray(a, np.float32)

    ref, hist, sim = make(), make(), make()
    gidx = _doy(T)
    q = ((np.arange(20) + 0.5) / 20).astype(np.float32)
    af, hq = hip.eqm_train(ref, hist, gidx, 365, 31, q, "+")
    eaf, ehq = oracle.eqm_train(ref, hist, gidx, 365, 31, q, "+")
    assert_close(hq, ehq, rtol=0.0, what="hist_q")
    assert_close(af, eaf, rtol=0.0, what="af")
    for rw in (False, True):
        scen, sim_q = hip.qdm_adjust(sim, eaf, q, gidx, 31, rw, "nearest", "constant", "+")
        escen, esim_q = oracle.qdm_adjust(sim, eaf, q, gidx, 31, rw, "nearest", "constant", "+")
        assert_close(sim_q, esim_q, rtol=0.0, what=f"sim_q rw={rw}")
        assert_close(scen, escen, rtol=0.0, what=f"scen rw={rw}")
    if kind != "zero_most_wide":  # (a group mean next to zero beside huge values: the guards hand such cells to the generic kernel anyway)
        af2, hq2, sc2 = hip.dqm_train(ref + np.float32(5.0), hist + np.float32(5.0), gidx, 365, 31, q, "*")
        eaf2, ehq2, esc2 = oracle.dqm_train(ref + np.float32(5.0), hist + np.float32(5.0), gidx, 365, 31, q, "*")
        assert_close(sc2, esc2, rtol=2e-6, what="scaling")
        assert_close(hq2, ehq2, rtol=2e-6, what="dqm hist_q")
