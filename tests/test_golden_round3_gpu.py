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
