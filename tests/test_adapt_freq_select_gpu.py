"""adapt_freq without a row limit (reference _processing.py:75-142; oracle: xsdba_oracle_adapt_freq): what the LDS does not sort —
the ref row of pth, the sub-threshold (key, tag) pairs of the ranks — is decided by selection over the row in global memory
(preproc.hpp block_select2; the default everywhere since it is no slower).  Against the oracle bit for bit, and against the LDS
sorts where both apply (XSDBA_HIP_AF_SELECT=0: the LDS sorts wherever they fit): windows, months, one group, NaNs, heavy ties (dry days are all
0.0: the tags decide), nothing under the threshold, a single valid value, rows beyond the LDS sort (21 900 float64 steps; more
than 16 384 values under the threshold) and beyond 65 535 values (slots beyond the 16 bits of the LDS tags)."""
import os

import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu


def _doy_month(T):
    d0 = np.arange(T) % 365
    cum = np.concatenate([[0], np.cumsum([31, 28, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31])])
    return d0.astype(np.int32), (np.searchsorted(cum, d0, side="right") - 1).astype(np.int32)


def _pair(C, T, rng, dtype, dry_sim, dry_ref):
    """precipitation-like: sim drier than ref in most cells (dP0 > 0: something to adapt), wetter in one"""
    ref = rng.gamma(0.8, 4.0, (C, T))
    sim = rng.gamma(0.8, 4.0, (C, T))
    ref[rng.random((C, T)) < dry_ref] = 0.0
    sim[rng.random((C, T)) < dry_sim] = 0.0
    sim[1] = np.where(rng.random(T) < 0.1, 0.0, sim[1] + 2.0)      # wetter than ref: dP0 <= 0, untouched
    sim[2] = np.round(sim[2], 1)                                   # ties above and below the threshold
    sim[3, rng.random(T) < 0.1] = np.nan
    ref[3, rng.random(T) < 0.1] = np.nan
    if C > 4:
        sim[4] = sim[4] + 5.0                                      # nothing under the threshold ...
        ref[4] = ref[4] + 5.0
        ref[4, ::7] = 0.0                                          # ... (P0_hist = 0: dP0 NaN)
    if C > 5:
        sim[5] = np.nan
        sim[5, T // 2] = 0.5                                       # a single valid value
    return sim.astype(dtype), ref.astype(dtype)


def _check(hip, sim, ref, gidx, G, W, select, seed=5):
    if not select:
        os.environ["XSDBA_HIP_AF_SELECT"] = "0"
    try:
        h = hip.adapt_freq(sim, ref, gidx, G, W, 1.0, seed=seed, first_cell=3)
        e = oracle.adapt_freq(sim, ref, gidx, G, W, 1.0, seed=seed, first_cell=3)
        for a, b, name in zip(h, e, ("sim_ad", "pth", "dP0", "P0_ref", "P0_hist")):
            np.testing.assert_array_equal(np.asarray(a), np.asarray(b), err_msg=name)
        # the adjust side (groups without window) on another series, with the trained tables
        if W == 1:
            sim2 = np.roll(np.asarray(sim), 17, axis=1).copy()
            h2 = hip.adapt_freq(sim2, None, gidx, G, 1, 1.0, seed=seed + 1, first_cell=3, P0_ref=e[3], P0_hist=e[4], pth=e[1])
            e2 = oracle.adapt_freq(sim2, None, gidx, G, 1, 1.0, seed=seed + 1, first_cell=3, P0_ref=e[3], P0_hist=e[4], pth=e[1])
            np.testing.assert_array_equal(np.asarray(h2[0]), np.asarray(e2[0]))
            np.testing.assert_array_equal(np.asarray(h2[2]), np.asarray(e2[2]))
        return np.asarray(h[0])
    finally:
        os.environ.pop("XSDBA_HIP_AF_SELECT", None)


@pytest.mark.parametrize("gname", ["time", "month", "doy31", "doy1"])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_selection_equals_the_lds_sorts_and_the_oracle(gname, dtype):
    from xsdba_amd import hip

    T = 3650
    rng = np.random.default_rng(len(gname) + (dtype == np.float64))
    doy, mon = _doy_month(T)
    gidx, G, W = {"time": (np.zeros(T, np.int32), 1, 1), "month": (mon, 12, 1), "doy31": (doy, 365, 31), "doy1": (doy, 365, 1)}[gname]
    sim, ref = _pair(6, T, rng, dtype, 0.5, 0.3)
    a = _check(hip, sim, ref, gidx, G, W, select=False)
    b = _check(hip, sim, ref, gidx, G, W, select=True)
    np.testing.assert_array_equal(a, b)
    assert (a != sim)[~np.isnan(sim)].any()   # (something was adapted)


@pytest.mark.parametrize("T,dtype,dry", [(21900, np.float64, 0.5),    # the ref row beyond the LDS sort (16 384 float64 keys)
                                          (21900, np.float32, 0.85),   # more than 16 384 values under the threshold
                                          (40000, np.float32, 0.5),    # the ref row beyond 32 768 float32 keys
                                          (70001, np.float32, 0.6),    # beyond 65 535 values: 32-bit slots
                                          (70001, np.float64, 0.6)])
def test_rows_beyond_the_lds(T, dtype, dry):
    from xsdba_amd import hip

    rng = np.random.default_rng(T)
    sim, ref = _pair(4, T, rng, dtype, dry, 0.3)
    a = _check(hip, sim, ref, np.zeros(T, np.int32), 1, 1, select=False)   # (the LDS sorts for whatever part fits them)
    b = _check(hip, sim, ref, np.zeros(T, np.int32), 1, 1, select=True)
    np.testing.assert_array_equal(a, b)
    assert (a != sim)[~np.isnan(sim)].any()


def test_months_of_a_long_hourly_series():
    """12 groups of ~ 22 000 hourly steps each (30 years): rows beyond the float64 sort, some with more sub-threshold values than
    the pair sort takes, some not — both ways in one launch"""
    from xsdba_amd import hip

    T = 24 * 365 * 30
    rng = np.random.default_rng(1)
    _, mon = _doy_month(T // 24)
    gidx = np.repeat(np.tile(mon[:365], 30), 24).astype(np.int32)
    sim, ref = _pair(4, T, rng, np.float64, 0.3, 0.2)
    m = np.isin(gidx, (0, 1, 2))
    sim[0, m & (rng.random(T) < 0.5)] = 0.0   # winter months of one cell: > 8 192 values under the threshold
    _check(hip, sim, ref, gidx, 12, 1, select=False)
    _check(hip, sim, ref, gidx, 12, 1, select=True)
