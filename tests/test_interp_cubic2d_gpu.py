"""Grouped interp="cubic" (qm_adjust -> interp_on_quantiles -> _interp_on_quantiles_2d with method="cubic", utils.py:380-400,
476-513): scipy.griddata(method="cubic") = CloughTocher2DInterpolator on the Delaunay triangulation of a cell's table nodes.
The device path (csrc/delaunay.hpp: dt_gradients + dt_eval_ct, k_lin2d_build / k_lin2d_adjust) is checked against griddata
itself, composed exactly like the reference in oracle.interp_on_quantiles_2d.  The reference's own test of this path
(tests/test_utils.py:116-146) only asserts that no NaN comes out (re-expressed in tests/test_api_gpu.py)."""
from __future__ import annotations

import numpy as np
import pytest

import oracle
import xsdba_amd as xs
from xsdba_amd import hip

from test_interp_linear_gpu import check, cube, gen, wrap

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    hip.default_context(0)


def expected_cubic(sim, af, hq, gcoord, newg, extrap, kind, trend=None):
    out = np.empty_like(sim)
    for c in range(sim.shape[0]):
        x = sim[c]
        if trend is not None:
            x = x - trend[c] if kind == "+" else x * (np.float32(1) / trend[c])
        oldg = np.repeat(gcoord[:, None], af.shape[2], 1)
        a = oracle.interp_on_quantiles_2d(x, newg, wrap(hq[c]), wrap(af[c]), oldg, "cubic", extrap)
        y = x + a if kind == "+" else x * a
        if trend is not None:
            y = y + trend[c] if kind == "+" else y * trend[c]
        out[c] = y
    return out


@pytest.mark.parametrize("group", ["time.month", "time.season"])
@pytest.mark.parametrize("kind", ["+", "*"])
@pytest.mark.parametrize("extrap", ["constant", "nan"])
def test_eqm_grouped_cubic_matches_griddata(group, kind, extrap):
    rng = np.random.default_rng(31)
    C, T, Q = 3, 6 * 365, 20
    ref, hist, sim = gen(rng, C, T)
    sim[1, rng.random(T) < 0.02] = np.nan
    dims = ("lat", "time")
    gr = xs.Grouper(group)
    eqm = xs.EmpiricalQuantileMapping.train(cube(ref, dims), cube(hist, dims), group=gr, kind=kind, nquantiles=Q)
    p = eqm.adjust(cube(sim, dims), interp="cubic", extrapolation=extrap).transpose(*dims).values
    prop = gr.prop
    af = np.ascontiguousarray(eqm.ds.af.transpose("lat", prop, "quantiles").values)
    hq = np.ascontiguousarray(eqm.ds.hist_q.transpose("lat", prop, "quantiles").values)
    crd = np.arange(1, 13.0) if prop == "month" else np.arange(4.0)
    gcoord = np.concatenate([[crd[0] - 1], crd, [crd[-1] + 1]])
    newg = np.asarray(gr.get_index(cube(sim, dims), interp=True).values, np.float64)
    check(p, expected_cubic(sim, af, hq, gcoord, newg, extrap, kind))
    assert p.dtype == np.float32
    # and it is not the piecewise-linear interpolant
    lin = eqm.adjust(cube(sim, dims), interp="linear", extrapolation=extrap).transpose(*dims).values
    both = np.isfinite(lin) & np.isfinite(p)
    assert np.abs(lin[both] - p[both]).max() > 1e-3


def test_grouped_cubic_with_nan_nodes_and_float64():
    rng = np.random.default_rng(32)
    C, T, Q = 2, 5 * 365, 15
    ref, hist, sim = (a.astype(np.float64) for a in gen(rng, C, T))
    dims = ("lat", "time")
    gr = xs.Grouper("time.month")
    eqm = xs.EmpiricalQuantileMapping.train(cube(ref, dims), cube(hist, dims), group=gr, nquantiles=Q)
    af = eqm.ds.af.transpose("lat", "month", "quantiles").values.copy()
    hq = eqm.ds.hist_q.transpose("lat", "month", "quantiles").values.copy()
    af[0, 4, :] = np.nan
    af[1, 2, 3] = np.nan
    hq[1, 7, 0] = np.nan
    gcoord = np.arange(0, 14.0)
    newg = np.asarray(gr.get_index(cube(sim, dims), interp=True).values, np.float64)
    got = hip.qm_adjust_linear2d(sim, af, hq, gcoord, newg, "constant", "+", cubic=True)
    exp = np.empty_like(sim)
    for c in range(C):
        a = oracle.interp_on_quantiles_2d(sim[c], newg, wrap(hq[c]), wrap(af[c]), np.repeat(gcoord[:, None], Q, 1), "cubic", "constant")
        exp[c] = sim[c] + a
    nan_mismatch = np.isnan(got) != np.isnan(exp)
    both = ~np.isnan(got) & ~np.isnan(exp)
    assert nan_mismatch.sum() <= 2e-3 * got.size
    # the gradients are an iterative solution (tol 1e-6): both sides stop at the same sweep, rounding differs
    assert (~np.isclose(got[both], exp[both], rtol=1e-8, atol=1e-8)).sum() <= 2e-3 * got.size


def test_eqm_dayofyear_cubic():
    """day-of-year tables: 367 x Q nodes per cell (triangulation and gradients stay in global memory).  Windowed rows share
    order statistics, so a fifth of the nodes sit exactly above a node of the next row: rectangles, i.e. cocircular
    quadruples whose diagonal is the triangulator's choice (Qhull's is arbitrary) and feeds into the gradients around it.
    (1) with the ties broken by a jitter of the abscissae the device path reproduces griddata; (2) on the tied table the two
    stay an order of magnitude closer to each other than the cubic interpolant is to the linear one."""
    rng = np.random.default_rng(33)
    C, T, Q = 2, 6 * 365, 10
    ref, hist, sim = gen(rng, C, T)
    dims = ("lat", "time")
    gr = xs.Grouper("time.dayofyear", 31)
    eqm = xs.EmpiricalQuantileMapping.train(cube(ref, dims), cube(hist, dims), group=gr, nquantiles=Q)
    af = np.ascontiguousarray(eqm.ds.af.transpose("lat", "dayofyear", "quantiles").values)
    hq = np.ascontiguousarray(eqm.ds.hist_q.transpose("lat", "dayofyear", "quantiles").values)
    gcoord = np.arange(0, 367.0)
    newg = np.asarray(gr.get_index(cube(sim, dims), interp=True).values, np.float64)
    # (1)
    hqj = np.sort(hq.astype(np.float64) + rng.uniform(0, 0.02, hq.shape), axis=-1)  # (float64: no accidental new ties)
    assert not (hqj[:, 1:, :, None] == hqj[:, :-1, None, :]).any()
    sim64, af64 = sim.astype(np.float64), af.astype(np.float64)
    got = hip.qm_adjust_linear2d(sim64, af64, hqj, gcoord, newg, "constant", "+", cubic=True)
    check(got, expected_cubic(sim64, af64, hqj, gcoord, newg, "constant", "+"), frac_tol=5e-3)
    # (2)
    p = eqm.adjust(cube(sim, dims), interp="cubic").transpose(*dims).values
    lin = eqm.adjust(cube(sim, dims), interp="linear").transpose(*dims).values
    exp = expected_cubic(sim, af, hq, gcoord, newg, "constant", "+")
    assert np.isfinite(p).all()
    d, dl = np.abs(p - exp), np.abs(p - lin)
    assert np.median(d) < 1e-3 and np.quantile(d, 0.99) < 0.2 * np.quantile(dl, 0.99) and d.max() < 0.2 * dl.max()


def test_qdm_grouped_cubic_runs_on_the_regular_grid():
    """QDM looks sim_q up in the regular (quantile, group) grid, whose cells are cocircular quadruples: the diagonals are
    the triangulator's choice (see test_interp_linear_gpu), so only the node values and smoothness are asserted"""
    rng = np.random.default_rng(35)
    ref, hist, sim = gen(rng, 2, 5 * 365)
    dims = ("lat", "time")
    gr = xs.Grouper("time.month")
    qdm = xs.QuantileDeltaMapping.train(cube(ref, dims), cube(hist, dims), group=gr, nquantiles=10)
    lin = qdm.adjust(cube(sim, dims), interp="linear").transpose(*dims).values
    cub = qdm.adjust(cube(sim, dims), interp="cubic").transpose(*dims).values
    assert np.isfinite(cub).all()
    af = qdm.ds.af.values
    assert np.abs(cub - lin).max() <= 0.5 * (af.max() - af.min())
    assert np.median(np.abs(cub - lin)) < 0.05 * (af.max() - af.min())


@pytest.mark.parametrize("kind", ["+", "*"])
def test_dqm_grouped_cubic(kind):
    """dqm_adjust with interp="cubic" on monthly groups (_adjustment.py:748-773): the scaling table shares `lat` with the
    series, so utils.broadcast falls back to linear with a warning (utils.py:228-234); then fit / detrend / Clough-Tocher
    mapping / retrend"""
    rng = np.random.default_rng(37)
    C, T, Q = 2, 6 * 365, 12
    ref, hist, sim = gen(rng, C, T)
    sim = (sim + np.linspace(0, 3, T)[None]).astype(np.float32)
    dims = ("lat", "time")
    gr = xs.Grouper("time.month")
    with pytest.warns(UserWarning):
        dqm = xs.DetrendedQuantileMapping.train(cube(ref, dims), cube(hist, dims), group=gr, kind=kind, nquantiles=Q)
    from xsdba_amd._adjustment import _group_interp_nodes, dqm_adjust

    with pytest.warns(UserWarning, match="not cubic. Using linear"):
        out = dqm_adjust(dqm.ds.assign(sim=cube(sim, dims)), group=gr, interp="cubic", kind=kind, extrapolation="constant", detrend=1)
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
    exp = expected_cubic(scaled, af, hq, gcoord, newg, "constant", kind, trend=trend.astype(np.float32))
    got = out.scen.transpose(*dims).values
    both = ~np.isnan(got) & ~np.isnan(exp)
    assert (np.isnan(got) != np.isnan(exp)).sum() <= 2e-3 * got.size
    assert (~np.isclose(got[both], exp[both], rtol=3e-5, atol=3e-4)).sum() <= 2e-3 * got.size


@pytest.mark.parametrize("kind", ["+", "*"])
def test_scaling_cubic_on_a_lone_series(kind):
    """Scaling.adjust(interp="cubic") on a series without other dimensions: utils.broadcast interpolates the monthly factors
    with xarray's .interp(method="cubic") = scipy's interp1d(kind="cubic") through the table wrapped with its cyclic rows
    (utils.py:222-246)"""
    from scipy.interpolate import interp1d

    rng = np.random.default_rng(38)
    T = 4 * 365
    ref, hist, sim = (a[0] + (5 if kind == "*" else 0) for a in gen(rng, 1, T))
    dims = ("time",)
    sc = xs.Scaling.train(cube(ref, dims), cube(hist, dims), group="time.month", kind=kind)
    got = sc.adjust(cube(sim, dims), interp="cubic").values
    af = sc.ds.af.values.astype(np.float64)
    i = xs.Grouper("time.month").get_index(cube(sim, dims), interp=True).values
    fac = interp1d(np.arange(14.0), np.concatenate([af[-1:], af, af[:1]]), kind="cubic")(i).astype(np.float32)
    exp = sim + fac if kind == "+" else sim * fac
    np.testing.assert_allclose(got, exp, rtol=2e-6)
    lin = sc.adjust(cube(sim, dims), interp="linear").values
    assert np.abs(lin - got).max() > 1e-4


def test_loci_cubic_falls_back_to_linear_on_a_grid():
    """utils.broadcast (utils.py:228-234): with a dimension shared between the tables and the series the reference warns and
    interpolates linearly"""
    rng = np.random.default_rng(39)
    ref, hist, sim = (np.abs(a - 280).astype(np.float32) for a in gen(rng, 2, 4 * 365))
    dims = ("lat", "time")
    loci = xs.LOCI.train(cube(ref, dims, "mm/d"), cube(hist, dims, "mm/d"), group="time.month", thresh="1 mm/d")
    lin = loci.adjust(cube(sim, dims, "mm/d"), interp="linear").values
    with pytest.warns(UserWarning, match="not cubic. Using linear"):
        cub = loci.adjust(cube(sim, dims, "mm/d"), interp="cubic").values
    np.testing.assert_array_equal(cub, lin)


def test_loci_cubic_on_a_lone_series():
    """LOCI.adjust(interp="cubic") without other dimensions (_adjustment.py:930-933): factor and threshold of every time step
    from xarray's .interp(method="cubic") = scipy's interp1d(kind="cubic") through the tables wrapped with their cyclic rows"""
    from scipy.interpolate import interp1d

    rng = np.random.default_rng(40)
    T = 4 * 365
    ref, hist, sim = (np.abs(a[0] - 280).astype(np.float32) for a in gen(rng, 1, T))
    dims = ("time",)
    loci = xs.LOCI.train(cube(ref, dims, "mm/d"), cube(hist, dims, "mm/d"), group="time.month", thresh="1 mm/d")
    got = loci.adjust(cube(sim, dims, "mm/d"), interp="cubic").values
    i = xs.Grouper("time.month").get_index(cube(sim, dims), interp=True).values
    w = lambda a: np.concatenate([a[-1:], a, a[:1]]).astype(np.float64)  # noqa: E731
    fac = interp1d(np.arange(14.0), w(loci.ds.af.values), kind="cubic")(i).astype(np.float32)
    sth = interp1d(np.arange(14.0), w(loci.ds.hist_thresh.values), kind="cubic")(i).astype(np.float32)
    exp = np.clip(fac * (sim - sth) + np.float32(1), 0, None)
    np.testing.assert_allclose(got, exp, rtol=3e-6, atol=1e-6)
    assert np.abs(got - loci.adjust(cube(sim, dims, "mm/d"), interp="linear").values).max() > 1e-4
