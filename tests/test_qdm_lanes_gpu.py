"""k_qdm_lanes (qdm_adjust over exact groups of <= 32 members: a lane per group, members sorted in registers) against the
oracle (reference _adjustment.py:783-886 / utils.py:573-638 restated) and against the generic kernel (XSDBA_HIP_QDM=count),
bit for bit: ties anywhere and at the extremes (zero-inflated, capped, constant data), NaNs in sim, NaN factors (those
cells fall back), -0.0, leap days (a 366th group with few members), time steps without a group, extrapolation "nan"."""
import os

import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu


def _doy_index(T, leap=False):
    if not leap:
        return (np.arange(T) % 365).astype(np.int32), 365
    g = np.empty(T, np.int32)
    t, year = 0, 0
    while t < T:
        n = 366 if year % 4 == 0 else 365
        m = min(n, T - t)
        g[t:t + m] = np.arange(m)
        t += m
        year += 1
    return g, 366


def _case(name, C, T, seed):
    rng = np.random.default_rng(seed)
    gidx, G = _doy_index(T, leap=name == "leap")
    doy = gidx.astype(np.float64)
    sim = (5.0 + 3.0 * np.sin(2 * np.pi * doy / 365.0)[None, :] + rng.gamma(2.0, 2.0, (C, T))).astype(np.float32)
    if name == "zero_inflated":
        sim[rng.random((C, T)) < 0.45] = 0.0
    elif name == "capped":
        sim = np.minimum(sim, np.float32(9.0))
    elif name == "rounded":
        sim = np.round(sim)
    elif name == "constant":
        sim[:] = 3.0
        sim[1] = 0.0
    elif name == "nan_sim":
        sim[rng.random((C, T)) < 0.08] = np.nan
        sim[2] = np.nan
        sim[3, 10:] = np.nan
    elif name == "neg_zero":
        sim[rng.random((C, T)) < 0.2] = -0.0
        sim[rng.random((C, T)) < 0.2] = 0.0
        sim -= np.float32(0.0)
    elif name == "ungrouped":
        gidx = gidx.copy()
        gidx[rng.random(T) < 0.05] = -1
    return sim, gidx, G


@pytest.mark.parametrize("name", ["plain", "zero_inflated", "capped", "rounded", "constant", "nan_sim", "neg_zero", "leap", "ungrouped"])
@pytest.mark.parametrize("kind,extrap,Q", [("*", "constant", 20), ("+", "nan", 15), ("*", "constant", 50)])
def test_lanes_vs_oracle_and_generic(name, kind, extrap, Q):
    import zlib

    from xsdba_amd import hip

    C, T = 12, 10950 if name != "leap" else 10958
    sim, gidx, G = _case(name, C, T, zlib.crc32(f"{name}{kind}{Q}".encode()))
    rng = np.random.default_rng(7)
    q = ((np.arange(Q) + 0.5) / Q).astype(np.float32)
    af = (1.0 + 0.3 * rng.standard_normal((C, G, Q))).astype(np.float32)
    os.environ.pop("XSDBA_HIP_QDM", None)
    scen, _ = hip.qdm_adjust(sim, af, q, gidx, 31, False, "nearest", extrap, kind, want_sim_q=False)
    os.environ["XSDBA_HIP_QDM"] = "count"
    try:
        gen, _ = hip.qdm_adjust(sim, af, q, gidx, 31, False, "nearest", extrap, kind, want_sim_q=False)
    finally:
        os.environ.pop("XSDBA_HIP_QDM", None)
    exp, _ = oracle.qdm_adjust(sim, af, q, gidx, 31, False, "nearest", extrap, kind)
    np.testing.assert_array_equal(np.asarray(scen).view(np.uint32), np.asarray(gen).view(np.uint32))
    np.testing.assert_array_equal(np.asarray(scen), exp)


def test_lanes_nan_factors_fall_back():
    from xsdba_amd import hip

    C, T, Q = 9, 7300, 20
    sim, gidx, G = _case("plain", C, T, 5)
    rng = np.random.default_rng(3)
    q = ((np.arange(Q) + 0.5) / Q).astype(np.float32)
    af = (1.0 + 0.3 * rng.standard_normal((C, G, Q))).astype(np.float32)
    af[1, 17, 3] = np.nan          # one node
    af[4, 100:140] = np.nan        # whole rows: the 2-D nearest node lies in another group
    af[7] = np.nan                 # nothing valid
    scen, _ = hip.qdm_adjust(sim, af, q, gidx, 31, False, "nearest", "constant", "*", want_sim_q=False)
    exp, _ = oracle.qdm_adjust(sim, af, q, gidx, 31, False, "nearest", "constant", "*")
    np.testing.assert_array_equal(np.asarray(scen), exp)


@pytest.mark.parametrize("G,members", [(40, 20), (70, 32), (130, 7)])
def test_lanes_other_group_shapes(G, members):
    """groups need not be days of the year: G groups of `members` time steps each, interleaved"""
    from xsdba_amd import hip

    C, Q = 5, 12
    T = G * members
    rng = np.random.default_rng(G)
    gidx = rng.permutation(np.repeat(np.arange(G), members)).astype(np.int32)
    sim = rng.gamma(2.0, 2.0, (C, T)).astype(np.float32)
    sim[rng.random((C, T)) < 0.3] = 0.0
    q = ((np.arange(Q) + 0.5) / Q).astype(np.float32)
    af = (1.0 + 0.3 * rng.standard_normal((C, G, Q))).astype(np.float32)
    scen, _ = hip.qdm_adjust(sim, af, q, gidx, 1, False, "nearest", "constant", "*", want_sim_q=False)
    exp, _ = oracle.qdm_adjust(sim, af, q, gidx, 1, False, "nearest", "constant", "*")
    np.testing.assert_array_equal(np.asarray(scen), exp)


@pytest.mark.parametrize("Q", [1, 2, 7, 100, 254])
def test_lanes_quantile_counts(Q):
    """any number of nodes the one-byte node code can address (Q < 255); one node: every percent rank maps to it"""
    from xsdba_amd import hip

    C, T = 4, 3650
    sim, gidx, G = _case("zero_inflated", C, T, 100 + Q)
    rng = np.random.default_rng(Q)
    q = ((np.arange(Q) + 0.5) / Q).astype(np.float32)
    af = (1.0 + 0.3 * rng.standard_normal((C, G, Q))).astype(np.float32)
    for extrap in ("constant", "nan"):
        scen, _ = hip.qdm_adjust(sim, af, q, gidx, 31, False, "nearest", extrap, "*", want_sim_q=False)
        exp, _ = oracle.qdm_adjust(sim, af, q, gidx, 31, False, "nearest", extrap, "*")
        np.testing.assert_array_equal(np.asarray(scen), exp)


def test_lanes_tiny_and_ragged():
    """two groups of 32 members; 129 groups (three chunks of 64, the last one lane wide) with 1..9 members each"""
    from xsdba_amd import hip

    rng = np.random.default_rng(11)
    for G, sizes in ((2, [32, 32]), (129, list(rng.integers(1, 10, 129)))):
        gidx = rng.permutation(np.repeat(np.arange(G), sizes)).astype(np.int32)
        T, C, Q = gidx.size, 3, 9
        sim = np.round(rng.gamma(2.0, 2.0, (C, T)), 1).astype(np.float32)
        q = ((np.arange(Q) + 0.5) / Q).astype(np.float32)
        af = (1.0 + 0.3 * rng.standard_normal((C, G, Q))).astype(np.float32)
        scen, _ = hip.qdm_adjust(sim, af, q, gidx, 1, False, "nearest", "constant", "+", want_sim_q=False)
        exp, _ = oracle.qdm_adjust(sim, af, q, gidx, 1, False, "nearest", "constant", "+")
        np.testing.assert_array_equal(np.asarray(scen), exp)


@pytest.mark.parametrize("name", ["plain", "zero_inflated", "capped", "nan_sim", "leap", "ungrouped"])
@pytest.mark.parametrize("kind,extrap", [("*", "constant"), ("+", "nan")])
def test_rank_window_without_sim_q(name, kind, extrap):
    """rank_window=True without the sim_q output: k_qdm_cell maps 2r -> node through the per-cell table of the groups that
    share the middle group's (N, pmin, pmax) and evaluates the others directly — the same values as with sim_q requested
    (no table) and as the oracle"""
    import zlib

    from xsdba_amd import hip

    C, T, Q = 6, 10950 if name != "leap" else 10958, 20
    sim, gidx, G = _case(name, C, T, zlib.crc32(f"rw{name}{kind}".encode()))
    rng = np.random.default_rng(17)
    q = ((np.arange(Q) + 0.5) / Q).astype(np.float32)
    af = (1.0 + 0.3 * rng.standard_normal((C, G, Q))).astype(np.float32)
    scen, none = hip.qdm_adjust(sim, af, q, gidx, 31, True, "nearest", extrap, kind, want_sim_q=False)
    full, sim_q = hip.qdm_adjust(sim, af, q, gidx, 31, True, "nearest", extrap, kind, want_sim_q=True)
    exp, esq = oracle.qdm_adjust(sim, af, q, gidx, 31, True, "nearest", extrap, kind)
    assert none is None
    np.testing.assert_array_equal(np.asarray(scen).view(np.uint32), np.asarray(full).view(np.uint32))
    np.testing.assert_array_equal(np.asarray(scen), exp)
    np.testing.assert_array_equal(np.asarray(sim_q), esq)


@pytest.mark.parametrize("name", ["plain", "zero_inflated", "nan_sim", "neg_zero", "leap", "ungrouped"])
@pytest.mark.parametrize("kind,extrap", [("*", "constant"), ("+", "nan")])
def test_lanes_float64(name, kind, extrap):
    """the float64 instantiation (64-bit keys beside their payload) against the oracle and the generic kernel"""
    import zlib

    from xsdba_amd import hip

    C, T, Q = 7, 10950 if name != "leap" else 10958, 20
    sim32, gidx, G = _case(name, C, T, zlib.crc32(f"f64{name}{kind}".encode()))
    rng = np.random.default_rng(23)
    sim = sim32.astype(np.float64)
    ok = np.isfinite(sim) & (sim != 0)
    sim[ok] += 1e-9 * rng.standard_normal(int(ok.sum()))  # values no float32 holds
    q = (np.arange(Q) + 0.5) / Q
    af = 1.0 + 0.3 * rng.standard_normal((C, G, Q))
    os.environ.pop("XSDBA_HIP_QDM", None)
    scen, _ = hip.qdm_adjust(sim, af, q, gidx, 31, False, "nearest", extrap, kind, want_sim_q=False)
    os.environ["XSDBA_HIP_QDM"] = "count"
    try:
        gen, _ = hip.qdm_adjust(sim, af, q, gidx, 31, False, "nearest", extrap, kind, want_sim_q=False)
    finally:
        os.environ.pop("XSDBA_HIP_QDM", None)
    exp, _ = oracle.qdm_adjust(sim, af, q, gidx, 31, False, "nearest", extrap, kind)
    assert np.asarray(scen).dtype == np.float64
    np.testing.assert_array_equal(np.asarray(scen).view(np.uint64), np.asarray(gen).view(np.uint64))
    np.testing.assert_array_equal(np.asarray(scen), exp)
