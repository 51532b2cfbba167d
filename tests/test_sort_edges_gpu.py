"""The per-cell sort behind the windowed training kernels (train_cell.hpp: one 14-bit counting pass + as many stable nibble
passes as the occupied key range needs; the last stable pass also leaves rank[]): series whose keys span fewer than 14 bits (no
stable pass at all: rank[] comes from the fallback loop), exactly 14 + 4k bits, every bit of a float32, one value only, one
valid value only — EQM / DQM tables and QDM ranks bit for bit against the oracle, float32 and the sizes of all three EPT
instantiations."""
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
