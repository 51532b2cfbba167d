"""GPU parity (through the C ABI) against the golden vectors added in round 3 (tests/golden/make_golden.py round3): the
deterministic half of processing.jitter and _processing._adapt_freq.  The same checks pin the oracle on the CPU side
(tests/test_oracle_golden.py)."""
from __future__ import annotations

import numpy as np
import pytest

from preproc_golden_checks import check_adapt_freq, check_jitter

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from xsdba_amd import hip as h

    h.default_context(0)
    return h


def test_adapt_freq_deterministic_half(hip, golden):
    check_adapt_freq(hip, golden("preproc_deterministic"), refuses_multiplicity=True)


def test_jitter_bounds_and_masks(hip, golden):
    check_jitter(hip, golden("preproc_deterministic"))


def test_window_group_bookkeeping_hand_case_nonreducing(hip, golden):
    """rank and trend through the non-reducing path of Grouper.apply (window centre slot base.py:438-439, time order :442-443):
    the explicit-loop expectations of make_golden.py round3 (the reference's test layout, tests/test_base.py:122-166)"""
    d = golden("window_groups_handcase_nonreducing")
    x, W = d["x"][None, :], int(d["W"])
    r = np.asarray(hip.rank_groups(x, d["gidx"], 12, W, rank_window=True))
    assert np.array_equal(np.isnan(r[0]), np.isnan(d["rank_pct"]))
    np.testing.assert_allclose(r[0], d["rank_pct"], rtol=1e-14, atol=1e-15)
    tr = np.asarray(hip.polydetrend_trend(x, d["gidx"], 12, W, d["time_ns"], 1))
    np.testing.assert_allclose(tr[0], d["trend"], rtol=1e-12)
    # float32 series: the same expectations to float32 rounding
    x32 = x.astype(np.float32)
    r32 = np.asarray(hip.rank_groups(x32, d["gidx"], 12, W, rank_window=True))
    np.testing.assert_allclose(r32[0], d["rank_pct"], rtol=1e-14, atol=1e-15)  # (x holds quarters: exact in float32, same ranks)
    tr32 = np.asarray(hip.polydetrend_trend(x32, d["gidx"], 12, W, d["time_ns"], 1))
    np.testing.assert_allclose(tr32[0], d["trend"], rtol=2e-7)


def test_week_and_season_bookkeeping_hand_case(hip, golden):
    """round 4: seasonal / weekly rows on the device against the python-datetime hand case (make_golden.py round4)"""
    d = golden("week_season_handcase")
    x = d["x"][None, :]
    gs, gw = d["season_gidx"], d["week_gidx"]
    np.testing.assert_allclose(hip.group_means(x, gs, 4, 1)[0], d["season_plain"], rtol=1e-14)
    np.testing.assert_allclose(hip.group_means(x, gs, 4, 5)[0], d["season_win5"], rtol=1e-14)
    np.testing.assert_allclose(hip.group_means(x, gw, 53, 1)[0], d["week_plain"], rtol=1e-14, equal_nan=True)
    np.testing.assert_allclose(hip.group_means(x, gw, 53, 3)[0], d["week_win3"], rtol=1e-14, equal_nan=True)
    x32 = x.astype(np.float32)
    med = hip.quantile_groups(x32, gs, 4, 1, np.array([0.5], np.float32))[0, :, 0]
    np.testing.assert_allclose(med, d["season_median"], rtol=1e-6)
    # the weekly medians through the training kernel too (53 groups, one of them empty -> NaN)
    medw = hip.quantile_groups(x32, gw, 53, 1, np.array([0.5], np.float32))[0, :, 0]
    assert np.isnan(medw[52]) and not np.isnan(medw[:52]).any()
