"""interp="cubic" on one-group tables (utils._interp_on_quantiles_1D, utils.py:350-377 -> scipy interp1d(kind="cubic"), a
not-a-knot cubic spline) against scipy itself (oracle.interp_on_quantiles_1d), plus the reference's own expectations
(tests/test_utils.py:68-113: 2.95 between the nodes of a linear table, constant / NaN outside)."""
from __future__ import annotations

import numpy as np
import pytest

import oracle
import xsdba_amd as xs
from xsdba_amd import hip
from xsdba_amd import utils as u
from xsdba_amd.labeled import DataArray

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    hip.default_context(0)


def tseries(values, calendar="noleap"):
    return DataArray(np.asarray(values), ("time",), {"time": xs.date_range("2000-01-01", len(values), calendar).values}, {"units": "K"})


@pytest.mark.parametrize("extrap", ["constant", "nan"])
@pytest.mark.parametrize("kind", ["+", "*"])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_eqm_time_cubic_matches_interp1d(extrap, kind, dtype):
    rng = np.random.default_rng(31)
    C, T, Q = 3, 3000, 25
    ref = (280 + 5 * rng.standard_normal((C, T))).astype(dtype)
    hist = (282 + 6 * rng.standard_normal((C, T))).astype(dtype)
    sim = (283 + 7 * rng.standard_normal((C, T))).astype(dtype)
    sim[0, rng.random(T) < 0.02] = np.nan
    mk = lambda a: DataArray(a, ("lat", "time"), {"lat": np.arange(C), "time": xs.date_range("2000-01-01", T, "noleap").values}, {"units": "K"})  # noqa: E731
    eqm = xs.EmpiricalQuantileMapping.train(mk(ref), mk(hist), group="time", kind=kind, nquantiles=Q)
    p = eqm.adjust(mk(sim), interp="cubic", extrapolation=extrap).transpose("lat", "time").values
    af = eqm.ds.af.transpose("lat", "group", "quantiles").values[:, 0]
    hq = eqm.ds.hist_q.transpose("lat", "group", "quantiles").values[:, 0]
    for c in range(C):
        a = oracle.interp_on_quantiles_1d(sim[c], hq[c], af[c], "cubic", extrap)
        exp = sim[c] + a if kind == "+" else sim[c] * a
        np.testing.assert_allclose(p[c], exp, rtol=2e-6 if dtype == np.float32 else 1e-11, atol=1e-5 if dtype == np.float32 else 1e-10)
    assert p.dtype == dtype


def test_cubic_with_nan_nodes():
    rng = np.random.default_rng(32)
    T, Q = 2000, 12
    sim = (10 + 3 * rng.standard_normal((1, T))).astype(np.float64)
    hq = np.sort(10 + 3 * rng.standard_normal((1, 1, Q)), axis=-1)
    af = rng.standard_normal((1, 1, Q))
    af[0, 0, 3] = np.nan
    hq[0, 0, 8] = np.nan
    got = hip.qm_adjust(sim, af, hq, None, "cubic", "constant", "+")
    exp = sim[0] + oracle.interp_on_quantiles_1d(sim[0], hq[0, 0], af[0, 0], "cubic", "constant")
    np.testing.assert_allclose(got[0], exp, rtol=1e-11, atol=1e-10)


@pytest.mark.parametrize("extrap,expf,expl", [("constant", 4.4, 2.9), ("nan", np.nan, np.nan)])
def test_reference_interp_on_quantiles_constant_cubic(extrap, expf, expl):
    """reference tests/test_utils.py:68-113 with interp="cubic": (nearest 2.9 / linear 2.95 /) cubic 2.95"""
    quantiles = np.linspace(0, 1, num=25)
    xq = DataArray(np.linspace(205, 229, num=25), ("quantiles",), {"quantiles": quantiles})
    yq = DataArray(np.linspace(2, -2, num=25), ("quantiles",), {"quantiles": quantiles})
    newx = tseries(np.linspace(240, 200, num=41) - 0.5)
    out = u.interp_on_quantiles(newx, xq, yq, group="time", method="cubic", extrapolation=extrap).values
    if np.isnan(expf):
        assert np.isnan(out[0]) and np.isnan(out[-1])
    else:
        assert out[0] == pytest.approx(-2) and out[-1] == pytest.approx(2)
    np.testing.assert_allclose(out[21], 2 - (218.5 - 205) / 6, rtol=1e-6)  # the table is a straight line: the spline is too
    xq2 = xq.copy(); yq2 = yq.copy()
    xq2.values[[0, 1]] = np.nan
    yq2.values[[-1, -2]] = np.nan
    out2 = u.interp_on_quantiles(newx, xq2, yq2, group="time", method="cubic", extrapolation=extrap).values
    exp2 = oracle.interp_on_quantiles_1d(newx.values, xq2.values, yq2.values, "cubic", extrap)
    np.testing.assert_allclose(out2, exp2, rtol=1e-9, atol=1e-9)


def test_qdm_time_cubic():
    rng = np.random.default_rng(33)
    T, Q = 2500, 15
    ref = rng.gamma(2.0, 3.0, T).astype(np.float32)
    hist = rng.gamma(2.0, 4.0, T).astype(np.float32)
    sim = rng.gamma(2.0, 5.0, T).astype(np.float32)
    qdm = xs.QuantileDeltaMapping.train(tseries(ref), tseries(hist), group="time", kind="*", nquantiles=Q)
    with xs.set_options(extra_output=True):
        out = qdm.adjust(tseries(sim), interp="cubic")
    af = qdm.ds.af.transpose("group", "quantiles").values[0]
    a = oracle.interp_on_quantiles_1d(out.sim_q.values, qdm.ds.quantiles.values.astype(np.float32), af, "cubic", "constant")
    np.testing.assert_allclose(out.scen.values, sim * a, rtol=3e-6)


def test_grouped_cubic_takes_the_triangulated_path():
    """a grouped table goes through griddata's Clough-Tocher interpolant (tests/test_interp_cubic2d_gpu.py); with ref == hist
    every factor is zero and so is the interpolant"""
    rng = np.random.default_rng(34)
    x = tseries((280 + rng.standard_normal(1500)).astype(np.float32))
    eqm = xs.EmpiricalQuantileMapping.train(x, x, group="time.month", nquantiles=5)
    np.testing.assert_array_equal(eqm.adjust(x, interp="cubic").values, x.values)
