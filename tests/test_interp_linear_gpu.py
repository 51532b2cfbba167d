"""Grouped interp="linear" (the reference's documented way to smooth monthly factors: qm_adjust -> interp_on_quantiles ->
_interp_on_quantiles_2d, utils.py:380-400, 476-513) against scipy.griddata itself, composed exactly like the reference in
oracle.interp_on_quantiles_2d.  The reference's own test of this path (tests/test_utils.py:116-146) only asserts that
no NaN comes out; the values are pinned here."""
from __future__ import annotations

import numpy as np
import pytest

import oracle
import xsdba_amd as xs
from xsdba_amd import hip
from xsdba_amd.labeled import DataArray

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    hip.default_context(0)


def cube(values, dims, units="K", calendar="noleap", start="2000-01-01"):
    values = np.asarray(values)
    coords = {"time": xs.date_range(start, values.shape[dims.index("time")], calendar).values}
    for d, n in zip(dims, values.shape):
        if d != "time":
            coords[d] = np.arange(n)
    return DataArray(values, dims, coords, {"units": units})


def wrap(tab):  # add_cyclic_bounds (utils.py:284-314)
    return np.concatenate([tab[-1:], tab, tab[:1]], axis=0)


def expected(sim, af, hq, gcoord, newg, extrap, kind, trend=None):
    out = np.empty_like(sim)
    for c in range(sim.shape[0]):
        x = sim[c]
        if trend is not None:
            x = x - trend[c] if kind == "+" else x * (np.float32(1) / trend[c])
        oldg = np.repeat(gcoord[:, None], af.shape[2], 1)
        a = oracle.interp_on_quantiles_2d(x, newg, wrap(hq[c]), wrap(af[c]), oldg, "linear", extrap)
        y = x + a if kind == "+" else x * a
        if trend is not None:
            y = y + trend[c] if kind == "+" else y * trend[c]
        out[c] = y
    return out


def check(got, exp, frac_tol=2e-3):
    """the values agree to float32 rounding; a handful of elements may fall on the other side of the hull boundary or on a
    cocircular diagonal (Qhull's choice there is arbitrary)"""
    assert got.shape == exp.shape
    nan_mismatch = np.isnan(got) != np.isnan(exp)
    both = ~np.isnan(got) & ~np.isnan(exp)
    bad = ~np.isclose(got[both], exp[both], rtol=3e-6, atol=3e-6)
    assert (nan_mismatch.sum() + bad.sum()) <= frac_tol * got.size, (nan_mismatch.sum(), bad.sum(), got.size)


def gen(rng, C, T, seasonal=8.0):
    t = np.arange(T)
    cyc = seasonal * np.cos(2 * np.pi * (t % 365) / 365)
    ref = (280 - cyc + 3 * rng.standard_normal((C, T))).astype(np.float32)
    hist = (282 - 0.8 * cyc + 4 * rng.standard_normal((C, T))).astype(np.float32)
    sim = (283 - 0.8 * cyc + 5 * rng.standard_normal((C, T))).astype(np.float32)
    return ref, hist, sim


@pytest.mark.parametrize("group", ["time.month", "time.season"])
@pytest.mark.parametrize("kind", ["+", "*"])
@pytest.mark.parametrize("extrap", ["constant", "nan"])
def test_eqm_grouped_linear_matches_griddata(group, kind, extrap):
    rng = np.random.default_rng(21)
    C, T, Q = 3, 6 * 365, 20
    ref, hist, sim = gen(rng, C, T)
    sim[1, rng.random(T) < 0.02] = np.nan
    dims = ("lat", "time")
    gr = xs.Grouper(group)
    eqm = xs.EmpiricalQuantileMapping.train(cube(ref, dims), cube(hist, dims), group=gr, kind=kind, nquantiles=Q)
    p = eqm.adjust(cube(sim, dims), interp="linear", extrapolation=extrap).transpose(*dims).values
    prop = gr.prop
    af = np.ascontiguousarray(eqm.ds.af.transpose("lat", prop, "quantiles").values)
    hq = np.ascontiguousarray(eqm.ds.hist_q.transpose("lat", prop, "quantiles").values)
    G = af.shape[1]
    crd = np.arange(1, 13.0) if prop == "month" else np.arange(4.0)
    gcoord = np.concatenate([[crd[0] - 1], crd, [crd[-1] + 1]])
    newg = np.asarray(gr.get_index(cube(sim, dims), interp=True).values, np.float64)
    assert G == len(crd)
    check(p, expected(sim, af, hq, gcoord, newg, extrap, kind))
    assert p.dtype == np.float32


def test_eqm_grouped_linear_with_nan_nodes_and_float64():
    rng = np.random.default_rng(22)
    C, T, Q = 2, 5 * 365, 15
    ref, hist, sim = (a.astype(np.float64) for a in gen(rng, C, T))
    dims = ("lat", "time")
    gr = xs.Grouper("time.month")
    eqm = xs.EmpiricalQuantileMapping.train(cube(ref, dims), cube(hist, dims), group=gr, nquantiles=Q)
    # knock out nodes: a whole month of cell 0, scattered nodes of cell 1
    af = eqm.ds.af.transpose("lat", "month", "quantiles").values.copy()
    hq = eqm.ds.hist_q.transpose("lat", "month", "quantiles").values.copy()
    af[0, 4, :] = np.nan
    af[1, 2, 3] = np.nan
    hq[1, 7, 0] = np.nan
    gcoord = np.arange(0, 14.0)
    newg = np.asarray(gr.get_index(cube(sim, dims), interp=True).values, np.float64)
    got = hip.qm_adjust_linear2d(sim, af, hq, gcoord, newg, "constant", "+")
    exp = np.empty_like(sim)
    for c in range(C):
        a = oracle.interp_on_quantiles_2d(sim[c], newg, wrap(hq[c]), wrap(af[c]), np.repeat(gcoord[:, None], Q, 1), "linear", "constant")
        exp[c] = sim[c] + a
    nan_mismatch = np.isnan(got) != np.isnan(exp)
    both = ~np.isnan(got) & ~np.isnan(exp)
    assert nan_mismatch.sum() <= 2e-3 * got.size
    assert (~np.isclose(got[both], exp[both], rtol=1e-9, atol=1e-9)).sum() <= 2e-3 * got.size


def test_eqm_dayofyear_linear():
    """day-of-year tables: 367 x Q nodes per cell (the triangulation stays in global memory)"""
    rng = np.random.default_rng(23)
    C, T, Q = 2, 6 * 365, 10
    ref, hist, sim = gen(rng, C, T)
    dims = ("lat", "time")
    gr = xs.Grouper("time.dayofyear", 31)
    eqm = xs.EmpiricalQuantileMapping.train(cube(ref, dims), cube(hist, dims), group=gr, nquantiles=Q)
    p = eqm.adjust(cube(sim, dims), interp="linear").transpose(*dims).values
    af = np.ascontiguousarray(eqm.ds.af.transpose("lat", "dayofyear", "quantiles").values)
    hq = np.ascontiguousarray(eqm.ds.hist_q.transpose("lat", "dayofyear", "quantiles").values)
    gcoord = np.arange(0, 367.0)
    newg = np.asarray(gr.get_index(cube(sim, dims), interp=True).values, np.float64)
    check(p, expected(sim, af, hq, gcoord, newg, "constant", "+"))


@pytest.mark.parametrize("kind", ["+", "*"])
def test_dqm_grouped_linear(kind):
    """dqm_adjust with interp="linear" on monthly groups (_adjustment.py:748-773): the scaling is interpolated between the
    month centres too (utils.broadcast), then fit / detrend / grouped-linear mapping / retrend"""
    rng = np.random.default_rng(24)
    C, T, Q = 2, 6 * 365, 12
    ref, hist, sim = gen(rng, C, T)
    sim = (sim + np.linspace(0, 3, T)[None]).astype(np.float32)
    dims = ("lat", "time")
    gr = xs.Grouper("time.month")
    with pytest.warns(UserWarning):
        dqm = xs.DetrendedQuantileMapping.train(cube(ref, dims), cube(hist, dims), group=gr, kind=kind, nquantiles=Q)
    from xsdba_amd._adjustment import _group_interp_nodes, dqm_adjust

    out = dqm_adjust(dqm.ds.assign(sim=cube(sim, dims)), group=gr, interp="linear", kind=kind, extrapolation="constant", detrend=1)
    af = np.ascontiguousarray(dqm.ds.af.transpose("lat", "month", "quantiles").values)
    hq = np.ascontiguousarray(dqm.ds.hist_q.transpose("lat", "month", "quantiles").values)
    sc = np.ascontiguousarray(dqm.ds.scaling.transpose("lat", "month").values)
    gidx, G = gr.group_index(cube(sim, dims))
    g_lo, g_hi, frac = _group_interp_nodes(gr, cube(sim, dims), gidx, G, "linear")
    scaled = oracle.scaling_adjust(sim, sc, g_lo, kind, g_hi=g_hi, frac=frac)
    tnum = gr._time_index(cube(sim, dims)).numeric_ns
    trend = oracle.polydetrend_trend(scaled, gidx, G, 1, tnum, 1)
    np.testing.assert_allclose(out.trend.transpose(*dims).values, trend, rtol=3e-6)
    gcoord = np.arange(0, 14.0)
    newg = np.asarray(gr.get_index(cube(sim, dims), interp=True).values, np.float64)
    exp = expected(scaled, af, hq, gcoord, newg, "constant", kind, trend=trend.astype(np.float32))
    got = out.scen.transpose(*dims).values
    both = ~np.isnan(got) & ~np.isnan(exp)
    assert (np.isnan(got) != np.isnan(exp)).sum() <= 2e-3 * got.size
    assert (~np.isclose(got[both], exp[both], rtol=3e-5, atol=3e-4)).sum() <= 2e-3 * got.size


def test_qdm_grouped_linear_on_the_regular_grid():
    """QDM looks sim_q up in the regular (quantile, group) grid: every grid cell is a cocircular quadruple, so the diagonal
    that splits it is the triangulator's choice (Qhull's looks random).  Both are linear interpolants of the same nodes:
    they agree where the four corners are coplanar, and never differ by more than the cell's twist |f00 + f11 - f01 - f10|/2."""
    rng = np.random.default_rng(25)
    ref, hist, sim = gen(rng, 2, 5 * 365)
    dims = ("lat", "time")
    gr = xs.Grouper("time.month")
    Q = 10
    qdm = xs.QuantileDeltaMapping.train(cube(ref, dims), cube(hist, dims), group=gr, nquantiles=Q)
    with xs.set_options(extra_output=True):
        out = qdm.adjust(cube(sim, dims), interp="linear")
    got = out.scen.transpose(*dims).values - sim
    sim_q = out.sim_q.transpose(*dims).values
    af = qdm.ds.af.transpose("lat", "month", "quantiles").values
    q = qdm.ds.quantiles.values.astype(np.float32)
    gcoord = np.arange(0, 14.0)
    newg = np.asarray(gr.get_index(cube(sim, dims), interp=True).values, np.float64)
    for c in range(2):
        exp = oracle.interp_on_quantiles_2d(sim_q[c], newg, np.repeat(q[None], 14, 0), wrap(af[c]), np.repeat(gcoord[:, None], Q, 1), "linear", "constant")
        w = wrap(af[c]).astype(np.float64)
        twist = np.abs(w[:-1, :-1] + w[1:, 1:] - w[:-1, 1:] - w[1:, :-1]).max() / 2
        assert np.isfinite(got[c]).all() and np.isfinite(exp).all()
        assert np.abs(got[c] - exp).max() <= twist + 1e-5
        assert np.median(np.abs(got[c] - exp)) < 0.05 * twist + 1e-6


def test_reference_qdm_seasonal():
    """reference tests/test_adjustment.py:804-833 (TestQDM.test_seasonal): group="time.season", interp="linear" """
    from scipy.stats import uniform

    rng = np.random.default_rng(26)
    uu = rng.random(10000)
    x, y = uniform(loc=1, scale=1).ppf(uu), uniform(loc=2, scale=4).ppf(uu)
    t = xs.date_range("2000-01-01", uu.size, "standard").values
    mk = lambda a: DataArray(a.astype(np.float32), ("time",), {"time": t}, {"units": "K"})  # noqa: E731
    qdm = xs.QuantileDeltaMapping.train(mk(y), mk(x), kind="+", group="time.season", nquantiles=10)
    p = qdm.adjust(mk(x), interp="linear").values
    middle = (uu > 1e-2) & (uu < 0.99)
    np.testing.assert_array_almost_equal(p[middle], y[middle].astype(np.float32), 1)


def test_reference_test_utils_grouped_interp_has_no_nan():
    """reference tests/test_utils.py:116-146 (test_interp_on_quantiles_monthly): interpolating the monthly factors must
    give no NaN, for every method the path offers"""
    t = np.arange(5 * 365)
    doy = t % 365 + 1
    rng = np.random.default_rng(0)
    ref = (280 - 20 * np.cos(2 * np.pi * doy / 365) + 2 * rng.random(t.size) + 0.1 * t / 365).astype(np.float32)
    hist = (280 - 18 * np.cos(2 * np.pi * doy / 365) + 2 * rng.random(t.size) + 0.11 * t / 365).astype(np.float32)
    r, h = cube(ref, ("time",)), cube(hist, ("time",))
    eqm = xs.EmpiricalQuantileMapping.train(r, h, group="time.month", nquantiles=15)
    for interp in ("nearest", "linear"):
        assert not np.isnan(eqm.adjust(h, interp=interp).values).any()


def test_utils_interp_on_quantiles_grouped_linear():
    """utils.interp_on_quantiles(method="linear") with a monthly table returns the interpolated factors themselves"""
    from xsdba_amd import utils as u

    rng = np.random.default_rng(26)
    ref, hist, sim = gen(rng, 1, 4 * 365)
    r, h, s = (cube(a[0], ("time",)) for a in (ref, hist, sim))
    gr = xs.Grouper("time.month")
    eqm = xs.EmpiricalQuantileMapping.train(r, h, group=gr, nquantiles=12)
    af_t = u.interp_on_quantiles(s, eqm.ds.hist_q, eqm.ds.af, group=gr, method="linear", extrapolation="constant").values
    af = eqm.ds.af.transpose("month", "quantiles").values
    hq = eqm.ds.hist_q.transpose("month", "quantiles").values
    gcoord = np.arange(0, 14.0)
    newg = np.asarray(gr.get_index(s, interp=True).values, np.float64)
    exp = oracle.interp_on_quantiles_2d(sim[0], newg, wrap(hq), wrap(af), np.repeat(gcoord[:, None], 12, 1), "linear", "constant")
    ok = np.isclose(af_t, exp, rtol=3e-6, atol=3e-6)
    assert (~ok).sum() <= 2e-3 * ok.size
