"""The per-cell sorts are stable because returning LDS atomics are served in ascending lane order — a measured, undocumented
property (profiles/r04_atomic_order.txt).  Every context probes it at creation (csrc/order_probe.hpp); a context
whose probe fails computes with the generic kernels instead.  Here: the probe passes on this device, and a context forced into
the fallback (XSDBA_HIP_ATOMIC_ORDER=broken) gives the oracle's tables, ranks and adjusted series bit for bit without ever
launching a kernel that relies on the property."""
from __future__ import annotations

import numpy as np
import pytest

import oracle
from conftest import assert_close
from xsdba_amd import hip
from xsdba_amd._lib import Context

pytestmark = pytest.mark.gpu

SORT_KERNELS = {"k_train_cell", "k_train_cell3", "k_train_long", "k_train_xlong", "k_train_cell_f64", "k_train_cell3_f64", "k_qdm_cell"}


def test_probe_passes_on_this_device():
    assert hip.default_context(0).lane_ordered_atomics is True


@pytest.fixture()
def broken(monkeypatch):
    monkeypatch.setenv("XSDBA_HIP_ATOMIC_ORDER", "broken")
    ctx = Context(0)
    monkeypatch.delenv("XSDBA_HIP_ATOMIC_ORDER")
    assert ctx.lane_ordered_atomics is False
    ctx.profile_enable(True)
    yield ctx
    ctx.close()


@pytest.mark.parametrize("T", [3650, 10950, 21900])
def test_fallback_context_matches_the_oracle(broken, T):
    rng = np.random.default_rng(T)
    C = 3
    doy = (np.arange(T) % 365).astype(np.int32)
    q = ((np.arange(20) + 0.5) / 20).astype(np.float32)
    ref = (280 + 8 * rng.standard_normal((C, T))).astype(np.float32)
    hist = (281 + 7 * rng.standard_normal((C, T))).astype(np.float32)
    sim = (282 + 7 * rng.standard_normal((C, T))).astype(np.float32)
    hist[1, rng.random(T) < 0.05] = np.nan
    af, hq = hip.eqm_train(ref, hist, doy, 365, 31, q, "+", ctx=broken)
    eaf, ehq = oracle.eqm_train(ref, hist, doy, 365, 31, q, "+")
    assert_close(hq, ehq, rtol=0.0, what="hist_q")
    assert_close(af, eaf, rtol=0.0, what="af")
    if T <= 10950:
        for rw in (False, True):
            scen, sq = hip.qdm_adjust(sim, af, q, doy, 31, rw, "nearest", "constant", "+", ctx=broken)
            escen, esq = oracle.qdm_adjust(sim, eaf, q, doy, 31, rw, "nearest", "constant", "+")
            np.testing.assert_array_equal(sq, esq)
            np.testing.assert_array_equal(scen, escen)
        d64 = ref.astype(np.float64), hist.astype(np.float64)
        af64, hq64 = hip.eqm_train(*d64, doy, 365, 31, q.astype(np.float64), "+", ctx=broken)
        eaf64, ehq64 = oracle.eqm_train(*d64, doy, 365, 31, q.astype(np.float64), "+")
        assert_close(hq64, ehq64, rtol=0.0, what="hist_q f64")
    ran = set(broken.profile_read())
    assert ran and not (ran & SORT_KERNELS), ran
    assert "k_train_rows" in ran
