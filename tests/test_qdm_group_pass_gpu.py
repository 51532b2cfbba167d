"""QDM ranks over few un-windowed groups (time.month, time.season, ...) by the group-keyed stable pass of k_qdm_cell (C3)
against the oracle (reference _adjustment.py:783-886 / utils.py:573-638 restated) and against the bit-vector walk it replaces
(XSDBA_HIP_QDM_WALK=1), bit for bit: the three LDS configurations of the kernel, float32 / float64, ties anywhere and at the
extremes of a group (zero-inflated, capped, constant data, -0.0), NaNs, a group without a valid value in one cell, 2 .. 64
groups in calendar order or interleaved at random, groups of very different sizes."""
import os

import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu


def _groups(name, T, rng):
    if name == "month":
        cum = np.concatenate([[0], np.cumsum([31, 28, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31])])
        return (np.searchsorted(cum, np.arange(T) % 365, side="right") - 1).astype(np.int32), 12
    if name == "season":
        g, _ = _groups("month", T, rng)
        return ((g + 1) % 12 // 3).astype(np.int32), 4
    if name == "two":
        return (np.arange(T) % 2).astype(np.int32), 2
    if name == "random64":
        return rng.integers(0, 64, T).astype(np.int32), 64
    if name == "skewed":  # one group holds nearly everything, the others a handful each (one of them a single step)
        g = np.zeros(T, np.int32)
        g[rng.choice(T, 40, replace=False)] = rng.integers(1, 7, 40)
        g[g == 7] = 0
        g[T // 2] = 7
        return g, 8
    raise KeyError(name)


def _series(data, C, T, rng, dtype):
    sim = rng.gamma(2.0, 2.0, (C, T))
    if data == "zero_inflated":
        sim[rng.random((C, T)) < 0.5] = 0.0
    elif data == "capped":
        sim = np.minimum(sim, 5.0)
        sim[1] = np.maximum(sim[1], 2.0)
    elif data == "rounded":
        sim = np.round(sim)
    elif data == "constant":
        sim[:] = 3.0
        sim[1] = 0.0
    elif data == "neg_zero":
        sim[rng.random((C, T)) < 0.3] = -0.0
        sim[rng.random((C, T)) < 0.3] = 0.0
    sim = sim.astype(dtype)
    if data == "nan":
        sim[rng.random((C, T)) < 0.1] = np.nan
        sim[2] = np.nan
        sim[3, 5:] = np.nan
    return sim


def _same_bits(a, b):
    """bit-identical wherever a number stands; NaN where the other has NaN (the payload of a NaN is whatever operand the
    multiply-add propagated it from: not part of the result)"""
    a, b = np.asarray(a), np.asarray(b)
    u = np.uint32 if a.dtype == np.float32 else np.uint64
    na, nb = np.isnan(a), np.isnan(b)
    np.testing.assert_array_equal(na, nb)
    np.testing.assert_array_equal(np.where(na, 0, a.view(u)), np.where(nb, 0, b.view(u)))


def _both(hip, sim, af, q, gidx, interp, extrap, kind):
    os.environ.pop("XSDBA_HIP_QDM_WALK", None)
    scen, sim_q = hip.qdm_adjust(sim, af, q, gidx, 1, False, interp, extrap, kind)
    os.environ["XSDBA_HIP_QDM_WALK"] = "1"
    try:
        wscen, wsim_q = hip.qdm_adjust(sim, af, q, gidx, 1, False, interp, extrap, kind)
    finally:
        os.environ.pop("XSDBA_HIP_QDM_WALK", None)
    return np.asarray(scen), np.asarray(sim_q), np.asarray(wscen), np.asarray(wsim_q)


@pytest.mark.parametrize("T", [700, 4500, 10950])
@pytest.mark.parametrize("gname", ["month", "season", "two", "random64", "skewed"])
@pytest.mark.parametrize("data", ["plain", "zero_inflated", "capped", "rounded", "constant", "neg_zero", "nan"])
def test_group_pass_vs_oracle_and_walk(T, gname, data):
    import zlib

    from xsdba_amd import hip

    rng = np.random.default_rng(zlib.crc32(f"{T}{gname}{data}".encode()))
    C, Q = 5, 15
    gidx, G = _groups(gname, T, rng)
    sim = _series(data, C, T, rng, np.float32)
    if data == "nan":
        sim[4, gidx == G // 2] = np.nan    # the middle group (it seeds pass D's node table) without a valid value
        sim[0, gidx == 0] = np.nan         # ... and the first
        sim[1, gidx == G - 1] = np.nan     # ... and the last
    q = ((np.arange(Q) + 0.5) / Q).astype(np.float32)
    af = (1.0 + 0.3 * rng.standard_normal((C, G, Q))).astype(np.float32)
    for kind, extrap in (("*", "constant"), ("+", "nan")):
        scen, sim_q, wscen, wsim_q = _both(hip, sim, af, q, gidx, "nearest", extrap, kind)
        escen, esim_q = oracle.qdm_adjust(sim, af, q, gidx, 1, False, "nearest", extrap, kind)
        np.testing.assert_array_equal(sim_q, esim_q)
        np.testing.assert_array_equal(scen, escen)
        _same_bits(sim_q, wsim_q)
        _same_bits(scen, wscen)
    # no sim_q wanted: pass D takes the tabulated nodes of the middle group's triple
    os.environ.pop("XSDBA_HIP_QDM_WALK", None)
    s2, _ = hip.qdm_adjust(sim, af, q, gidx, 1, False, "nearest", "constant", "*", want_sim_q=False)
    np.testing.assert_array_equal(np.asarray(s2), oracle.qdm_adjust(sim, af, q, gidx, 1, False, "nearest", "constant", "*")[0])
    # the ranks alone
    r = hip.rank_groups(sim, gidx, G, 1, False)
    np.testing.assert_array_equal(np.asarray(r), oracle.rank_groups(sim, gidx, G, 1, False))


@pytest.mark.parametrize("T", [3000, 6000])
@pytest.mark.parametrize("gname", ["month", "random64"])
def test_group_pass_float64(T, gname):
    """31-bit range keys: equality inside a segment is decided on the doubles (thousands of distinct values on one key)"""
    from xsdba_amd import hip

    rng = np.random.default_rng(T)
    C, Q = 4, 15
    gidx, G = _groups(gname, T, rng)
    sim = _series("plain", C, T, rng, np.float64) * (1 + 1e-9 * rng.random((C, T)))
    sim[0] = np.round(sim[0] / 4) * 4
    sim[1, ::3] = 0.0
    sim[1, 1::6] = -0.0
    m = rng.random(T) < 0.4
    sim[2, m] = 50.0 + 1e-13 * rng.permutation(m.sum())
    sim[2, ::11] = 50.0
    sim[3, rng.random(T) < 0.05] = np.nan
    q = ((np.arange(Q) + 0.5) / Q).astype(np.float64)
    af = 1.0 + 0.3 * rng.standard_normal((C, G, Q))
    scen, sim_q, wscen, wsim_q = _both(hip, sim, af, q, gidx, "nearest", "constant", "*")
    escen, esim_q = oracle.qdm_adjust(sim, af, q, gidx, 1, False, "nearest", "constant", "*")
    assert scen.dtype == np.float64
    np.testing.assert_array_equal(sim_q, esim_q)
    np.testing.assert_array_equal(scen, escen)
    _same_bits(scen, wscen)
    _same_bits(sim_q, wsim_q)


def test_more_than_64_groups_and_ungrouped_steps_take_the_walk():
    """outside the pass's conditions (more groups than a wave has lanes for; steps without a group) nothing changes"""
    from xsdba_amd import hip

    rng = np.random.default_rng(9)
    C, T, Q = 3, 5000, 12
    sim = _series("zero_inflated", C, T, rng, np.float32)
    q = ((np.arange(Q) + 0.5) / Q).astype(np.float32)
    for G, holes in ((65, False), (12, True)):
        gidx = rng.integers(0, G, T).astype(np.int32)
        if holes:
            gidx[rng.random(T) < 0.1] = -1
        af = (1.0 + 0.3 * rng.standard_normal((C, G, Q))).astype(np.float32)
        scen, sim_q, wscen, wsim_q = _both(hip, sim, af, q, gidx, "nearest", "constant", "*")
        escen, esim_q = oracle.qdm_adjust(sim, af, q, gidx, 1, False, "nearest", "constant", "*")
        np.testing.assert_array_equal(sim_q, esim_q)
        np.testing.assert_array_equal(scen, escen)
        _same_bits(scen, wscen)
