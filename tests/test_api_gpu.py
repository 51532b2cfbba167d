"""The reference's own tests of the path, re-expressed against xsdba_amd (GPU): tests/test_adjustment.py
TestQM (:945-1031), TestDQM (:318-385), TestQDM (:700-744, :862-889), tests/test_utils.py
interp_on_quantiles (:68-146) and rank (:197-205), tests/test_nbutils.py (:10-34),
tests/test_detrending.py (:18-30).  They read like the originals: same inputs, same assertions."""
from __future__ import annotations

import warnings

import numpy as np
import pytest
from scipy.stats import norm, uniform

import xsdba_amd as xs
from xsdba_amd import nbutils as nbu
from xsdba_amd import utils as u
from xsdba_amd.labeled import DataArray
from xsdba_amd.utils import ADDITIVE, MULTIPLICATIVE, apply_correction, get_correction, invert

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    from xsdba_amd import hip

    hip.default_context(0)


@pytest.fixture
def random():
    return np.random.default_rng(seed=list(map(ord, "𝕽𝔞𝖓𝔡𝖔𝔪")))


def timelonlatseries(values, attrs=None, start="2000-01-01"):
    """reference testing/helpers.py:51-66 (time, lon, lat) builder on the standard calendar"""
    values = np.asarray(values)
    dims = ("time", "lon", "lat")[: values.ndim]
    coords = {"time": xs.date_range(start, values.shape[0], "standard").values}
    for d, n in zip(dims[1:], values.shape[1:]):
        coords[d] = np.arange(n)
    return DataArray(values, dims, coords, attrs or {})


def timeseries(values, start="2000-07-01", units=None, calendar=None):
    t = xs.date_range(start, len(values), calendar or "standard")
    return DataArray(np.asarray(values), ("time",), {"time": t.values}, {"units": units} if units else {})


MON_TRI = np.array(list(range(1, 7)) + list(range(7, 1, -1))) / 7


def mon_timelonlatseries(values, attrs):
    x = timelonlatseries(values, attrs)
    month = xs.Grouper("time.month").get_index(x).values
    factor = timelonlatseries(MON_TRI[month - 1], attrs)
    return DataArray(apply_correction(x, factor, attrs["kind"]).values, x.dims, x.coords, attrs)


class TestLoci:
    """reference tests/test_adjustment.py:156-203"""

    @pytest.mark.parametrize("group,dec", (["time", 2], ["time.month", 1]))
    def test_time_and_from_ds(self, group, dec, random):
        n = 10000
        x = uniform(loc=0, scale=3).ppf(random.random(n))
        hist = sim = timelonlatseries(x, attrs={"units": "kg m-2 s-1", "history": "ancient history."})
        y = x * 2
        thresh = 2
        ref_fit = timelonlatseries(np.where(y > thresh, y, 0.1), attrs={"units": "kg m-2 s-1"})
        ref = timelonlatseries(y, attrs={"units": "kg m-2 s-1"})
        loci = xs.LOCI.train(ref_fit, hist, group=group, thresh=f"{thresh} kg m-2 s-1")
        np.testing.assert_array_almost_equal(loci.ds.hist_thresh.values, 1, dec)
        np.testing.assert_array_almost_equal(loci.ds.af.values, 2, dec)
        p = loci.adjust(sim, interp="nearest")
        np.testing.assert_array_almost_equal(p.values, ref.values, dec)
        assert "Bias-adjusted with LOCI(" in p.attrs["history"] and p.attrs["history"].endswith("ancient history.")
        loci2 = xs.LOCI.from_dataset(loci.ds)
        np.testing.assert_array_equal(loci2.adjust(sim, interp="nearest").values, p.values)
        if group == "time":
            np.testing.assert_array_equal(loci.adjust(sim).values, p.values)  # default interp: one group, nothing to interpolate


class TestScaling:
    """reference tests/test_adjustment.py:206-258"""

    @pytest.mark.parametrize("kind,units", [(ADDITIVE, "K"), (MULTIPLICATIVE, "kg m-2 s-1")])
    def test_time(self, kind, units, random):
        n = 10000
        x = uniform(loc=2, scale=1).ppf(random.random(n))
        attrs = {"units": units, "kind": kind}
        hist = sim = timelonlatseries(x, attrs=attrs)
        ref = timelonlatseries(apply_correction(x, 2, kind), attrs=attrs)
        scaling = xs.Scaling.train(ref, hist, group="time", kind=kind)
        np.testing.assert_array_almost_equal(scaling.ds.af.values, 2)
        p = scaling.adjust(sim)
        np.testing.assert_array_almost_equal(p.values, ref.values)
        np.testing.assert_array_equal(scaling.adjust(sim, interp="linear").values, p.values)  # one group: nothing to interpolate

    @pytest.mark.parametrize("kind,units", [(ADDITIVE, "K"), (MULTIPLICATIVE, "kg m-2 s-1")])
    def test_mon_u(self, kind, units, random):
        n = 10000
        x = uniform(loc=2, scale=1).ppf(random.random(n))
        attrs = {"units": units, "kind": kind}
        hist = sim = timelonlatseries(x, attrs=attrs)
        ref = mon_timelonlatseries(apply_correction(x, 2, kind), attrs=attrs)
        scaling = xs.Scaling.train(ref, hist, group="time.month", kind=kind)
        expected = apply_correction(MON_TRI, 2, kind)
        np.testing.assert_array_almost_equal(scaling.ds.af.values, expected, 2 if kind == ADDITIVE else 1)
        p = scaling.adjust(sim)
        np.testing.assert_array_almost_equal(p.values, ref.values, 1)
        assert "Scaling(group=Grouper(name='time.month'), kind=" in scaling.ds.attrs["adj_params"]
        # interp="linear": the factor of a day is interpolated between the two nearest month centres (cyclic); checked
        # against a direct numpy evaluation of utils.broadcast's recipe (utils.py:222-246)
        pl = scaling.adjust(sim, interp="linear")
        af = scaling.ds.af.values.astype(np.float64)
        ext = np.concatenate([af[-1:], af, af[:1]])  # coordinates 0 .. 13
        i = xs.Grouper("time.month").get_index(sim, interp=True).values
        fac = np.interp(i, np.arange(14), ext)
        exp = sim.values + fac if kind == ADDITIVE else sim.values * fac
        np.testing.assert_allclose(pl.values, exp, rtol=1e-12)
        # interp="cubic" on a lone series: xarray's .interp(method="cubic") = scipy's interp1d(kind="cubic") over the same nodes
        from scipy.interpolate import interp1d

        pc = scaling.adjust(sim, interp="cubic")
        facc = interp1d(np.arange(14.0), ext, kind="cubic")(i)
        expc = sim.values + facc if kind == ADDITIVE else sim.values * facc
        np.testing.assert_allclose(pc.values, expc, rtol=1e-6)


class TestQM:
    @pytest.mark.parametrize("kind,units", [(ADDITIVE, "K"), (MULTIPLICATIVE, "kg m-2 s-1")])
    def test_quantiles(self, kind, units, random):
        u_ = random.random(10000)
        xd, yd = uniform(loc=10, scale=1), norm(loc=12, scale=1)
        x, y = xd.ppf(u_), yd.ppf(u_)
        hist = sim = timelonlatseries(x, attrs={"units": units})
        ref = timelonlatseries(y, attrs={"units": units})
        QM = xs.EmpiricalQuantileMapping.train(ref, hist, kind=kind, group="time", nquantiles=50)
        p = QM.adjust(sim, interp="linear")
        q = QM.ds.coords["quantiles"].values
        expected = get_correction(xd.ppf(q), yd.ppf(q), kind)[np.newaxis, :]
        np.testing.assert_array_almost_equal(QM.ds.af.values[:, 2:-2], expected[:, 2:-2], 1)
        middle = (u_ > 1e-2) * (u_ < 0.99)
        np.testing.assert_array_almost_equal(p.values[middle], ref.values[middle], 1)
        # trained dataset layout
        assert QM.ds.af.dims == ("group", "quantiles") and QM.ds.sizes["group"] == 1
        assert set(QM.ds.data_vars) == {"af", "hist_q", "hist_q_raw"} and np.isnan(QM.ds.hist_q_raw.values).all()
        assert QM.ds.attrs["group"] == "time" and QM.ds.attrs["group_window"] == 1
        assert "Bias-adjusted with EmpiricalQuantileMapping(" in p.attrs["history"] and p.attrs["units"] == units
        assert p.attrs["bias_adjustment"].endswith(".adjust(sim, interp='linear')")

    @pytest.mark.parametrize("kind,units", [(ADDITIVE, "K"), (MULTIPLICATIVE, "kg m-2 s-1")])
    def test_mon_u(self, kind, units, random):
        u_ = random.random(10000)
        xd, yd, noise = uniform(loc=2, scale=0.1), uniform(loc=4, scale=0.1), uniform(loc=0, scale=1e-7)
        x, y = xd.ppf(u_), yd.ppf(u_) + noise.ppf(u_)
        attrs = {"units": units, "kind": kind}
        hist = sim = timelonlatseries(x, attrs=attrs)
        ref = mon_timelonlatseries(y, attrs=attrs)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            QM = xs.EmpiricalQuantileMapping.train(ref, hist, kind=kind, group="time.month", nquantiles=5)
            p = QM.adjust(sim)
        mqm = QM.ds.af.mean(dim="quantiles")
        expected = apply_correction(MON_TRI, 2, kind)
        np.testing.assert_array_almost_equal(mqm.values, expected, 1)
        np.testing.assert_array_almost_equal(p.values, ref.values, 2)

    def test_grid_and_layouts(self, random):
        """(time, lon, lat) cubes go through without host transposes; result matches per-cell training"""
        T = 1460
        vals = random.random((T, 3, 4)).astype(np.float32) * 5 + 280
        t = xs.date_range("2001-01-01", T, "noleap").values
        mk = lambda v: DataArray(v, ("time", "lon", "lat"), {"time": t, "lon": np.arange(3), "lat": np.arange(4)}, {"units": "K"})  # noqa: E731
        ref, hist = mk(vals), mk(vals[::-1].copy())
        EQM = xs.EmpiricalQuantileMapping.train(ref, hist, group=xs.Grouper("time.dayofyear", window=31), nquantiles=10)
        assert EQM.ds.af.dims == ("lon", "lat", "dayofyear", "quantiles") and EQM.ds.af.dtype == np.float32
        scen = EQM.adjust(hist)
        assert set(scen.dims) == {"time", "lon", "lat"} and scen.dtype == np.float32
        one = xs.EmpiricalQuantileMapping.train(ref.isel(lon=1, lat=2), hist.isel(lon=1, lat=2),
                                                group=xs.Grouper("time.dayofyear", window=31), nquantiles=10)
        np.testing.assert_array_equal(one.ds.af.values, EQM.ds.af.values[1, 2])
        np.testing.assert_array_equal(one.adjust(hist.isel(lon=1, lat=2)).values, scen.transpose("lon", "lat", "time").values[1, 2])


class TestDQM:
    @pytest.mark.parametrize("kind,units", [(ADDITIVE, "K"), (MULTIPLICATIVE, "kg m-2 s-1")])
    def test_quantiles(self, kind, units, random):
        ns = 10000
        u_ = random.random(ns)
        xd, yd = uniform(loc=10, scale=1), norm(loc=12, scale=1)
        x, y = xd.ppf(u_), yd.ppf(u_)
        attrs = {"units": units, "kind": kind}
        hist = sim = timelonlatseries(x, attrs=attrs)
        ref = timelonlatseries(y, attrs=attrs)
        DQM = xs.DetrendedQuantileMapping.train(ref, hist, kind=kind, group="time", nquantiles=50)
        p = DQM.adjust(sim, interp="linear")
        q = DQM.ds.coords["quantiles"].values
        ex = apply_correction(xd.ppf(q), invert(xd.mean(), kind), kind)
        ey = apply_correction(yd.ppf(q), invert(yd.mean(), kind), kind)
        expected = get_correction(ex, ey, kind)
        np.testing.assert_array_almost_equal(DQM.ds.af.values[:, 2:-2], expected[np.newaxis, 2:-2], 1)
        # the reference masks on x (:363), which is vacuous for x in [10, 11]; the mapping is held
        # to 1 decimal on u in (0.03, 0.97) instead — its tails deviate by up to 0.18 in the oracle too
        middle = (u_ > 3e-2) * (u_ < 0.97)
        np.testing.assert_array_almost_equal(p.values[middle], ref.values[middle], 1)
        # with an actual trend in sim (:375-385)
        trend = timelonlatseries(np.linspace(-0.2, 0.2, ns) + (1 if kind == MULTIPLICATIVE else 0), attrs=attrs)
        sim3 = DataArray(apply_correction(sim, trend, kind).values, sim.dims, sim.coords, attrs)
        ref3 = apply_correction(ref, trend, kind)
        p3 = DQM.adjust(sim3, interp="linear")
        np.testing.assert_array_almost_equal(p3.values[middle], ref3.values[middle], 1)
        assert "scaling" in DQM.ds.data_vars and DQM.ds.scaling.dims == ("group",)

    def test_doy_window_360(self):
        # reference :455-491 test_360: doy window 31 on a 360-day calendar -> sizes {dayofyear: 360, quantiles: 50}
        rng = np.random.default_rng(0)
        T = 360 * 6
        t = xs.date_range("2000-01-01", T, "360_day").values
        mk = lambda v: DataArray(v, ("time",), {"time": t}, {"units": "K"})  # noqa: E731
        ref, hist = mk(rng.random(T) + 280), mk(rng.random(T) + 281)
        DQM = xs.DetrendedQuantileMapping.train(ref, hist, group=xs.Grouper("time.dayofyear", window=31), nquantiles=50)
        assert dict(DQM.ds.af.sizes) == {"dayofyear": 360, "quantiles": 50}
        scen = DQM.adjust(hist)
        assert np.isfinite(scen.values).all()

    def test_from_dataset_roundtrip(self, random):
        # reference :434-452: the trained dataset alone restores the object
        T = 1460
        t = xs.date_range("2001-01-01", T, "noleap").values
        mk = lambda v: DataArray(v, ("time",), {"time": t}, {"units": "K"})  # noqa: E731
        ref, hist = mk(random.random(T) + 280), mk(random.random(T) + 281)
        DQM = xs.DetrendedQuantileMapping.train(ref, hist, group="time.dayofyear", window=31, nquantiles=10)
        DQM2 = xs.DetrendedQuantileMapping.from_dataset(DQM.ds)
        assert repr(DQM2) == repr(DQM) and DQM2.group.window == 31
        np.testing.assert_array_equal(DQM.adjust(hist).values, DQM2.adjust(hist).values)


class TestQDM:
    @pytest.mark.parametrize("kind,units", [(ADDITIVE, "K"), (MULTIPLICATIVE, "kg m-2 s-1")])
    def test_quantiles(self, kind, units, random):
        u_ = random.random(10000)
        xd, yd = uniform(loc=1, scale=1), uniform(loc=2, scale=4)
        x, y = xd.ppf(u_), yd.ppf(u_)
        attrs = {"units": units, "kind": kind}
        hist = sim = timelonlatseries(x, attrs=attrs)
        ref = timelonlatseries(y, attrs=attrs)
        QDM = xs.QuantileDeltaMapping.train(ref.astype("float32"), hist.astype("float32"), kind=kind, group="time", nquantiles=10)
        p = QDM.adjust(sim.astype("float32"), interp="linear")
        q = QDM.ds.coords["quantiles"].values
        expected = get_correction(xd.ppf(q), yd.ppf(q), kind)[np.newaxis, :]
        np.testing.assert_array_almost_equal(QDM.ds.af.values, expected, 1)
        middle = (u_ > 1e-2) * (u_ < 0.99)
        np.testing.assert_array_almost_equal(p.values[middle], ref.values[middle], 1)
        assert QDM.ds.af.dtype == "float32" and p.dtype == "float32"  # dtype control (:743-744)

    def test_qdm_adjust_rank_window(self):
        def _daily_series(start, periods, offset=0):
            steps = np.arange(periods)
            values = 280 + np.sin(2 * np.pi * (steps % 365) / 365) + 0.01 * steps + offset
            return DataArray(values, ("time",), {"time": xs.date_range(start, periods, "noleap").values}, {"units": "K"})

        hist = _daily_series("2001-01-01", 2 * 365)
        ref = DataArray(hist.values + 2, hist.dims, hist.coords, {"units": "K"})
        sim = _daily_series("2003-01-01", 365, offset=1)
        group = xs.Grouper("time.dayofyear", window=31)
        QDM = xs.QuantileDeltaMapping.train(ref, hist, kind="+", group=group, nquantiles=10)
        with pytest.warns(DeprecationWarning, match="same window as used in the training.*will be deprecated in"):
            scen_default = QDM.adjust(sim)
        assert bool(np.isnan(scen_default.values).all())
        scen_window = QDM.adjust(sim, rank_window=True)
        assert bool(np.isfinite(scen_window.values).all())
        with xs.set_options(extra_output=True):
            out = QDM.adjust(sim, rank_window=True)
        assert bool(np.isfinite(out.sim_q.values).all())


@pytest.mark.parametrize("cls", ["EmpiricalQuantileMapping", "DetrendedQuantileMapping", "QuantileDeltaMapping"])
@pytest.mark.parametrize("group", ["time", "time.month"])
def test_max_tail_factor(random, cls, group):
    """reference tests/test_adjustment.py:597-647 (DQM), :891-941 (QDM), :1131-1181 (EQM): an outlier far above
    max_tail_factor x the largest hist quantile keeps its input value; without the option it is adjusted"""
    def series(start, n, hi, seed_vals=None):
        vals = random.uniform(0.001, hi, size=(n, 3))
        if seed_vals is not None:
            vals[0] = seed_vals
        t = xs.date_range(start, n, "standard").values
        return DataArray(vals, ("time", "lat"), {"time": t, "lat": np.arange(3)}, {"units": "mm d-1"})

    n_hist = 31 * 365 + 8  # 1990-01-01 .. 2020-12-31
    ref, hist = series("1990-01-01", n_hist, 20), series("1990-01-01", n_hist, 10)
    sim = series("2021-01-01", 30 * 365 + 7, 10, seed_vals=300)
    Adj = getattr(xs, cls)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        adj = Adj.train(ref, hist, kind="*", group=group)
        scen = adj.adjust(sim)
        assert float(scen.isel(time=0, lat=0).values) > 300
        assert bool(np.isnan(adj.ds.hist_q_raw.values).all())  # dummy variable without the option
        adj = Adj.train(ref, hist, kind="*", group=group, max_tail_factor=10)
        assert not np.isnan(adj.ds.hist_q_raw.values).any()
        scen10 = adj.adjust(sim)
    assert float(scen10.isel(time=0, lat=0).values) == 300
    # everything that is not masked is adjusted exactly as without the option
    a, b = scen.transpose("lat", "time").values, scen10.transpose("lat", "time").values
    assert (a != b).sum() == 3 and np.array_equal(a[:, 1:], b[:, 1:])
    assert "max_tail_factor=10" in adj.ds.attrs["adj_params"]


@pytest.mark.parametrize("interp", ["linear", "cubic"])
def test_max_tail_factor_with_interpolated_groups(random, interp):
    """_adjustment.py:648-658 with a grouped interp: the last raw hist quantile is broadcast between the month centres with the
    same method (utils.broadcast; 'cubic' falls back to 'linear' there because the table shares `lat` with the series)"""
    n = 6 * 365
    t = xs.date_range("2000-01-01", n, "noleap").values
    mk = lambda v: DataArray(v, ("time", "lat"), {"time": t, "lat": np.arange(2)}, {"units": "mm d-1"})  # noqa: E731
    season = (1.5 + np.cos(2 * np.pi * (np.arange(n) % 365) / 365))[:, None]
    ref, hist = mk(random.uniform(0.001, 20, (n, 2)) * season), mk(random.uniform(0.001, 10, (n, 2)) * season)
    simv = random.uniform(0.001, 10, (n, 2)) * season
    simv[::97] *= 4.0  # a good number of values around 2 x the largest hist quantile of their month
    sim = mk(simv)
    factor = 2.0
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        plain = xs.EmpiricalQuantileMapping.train(ref, hist, kind="*", group="time.month").adjust(sim, interp=interp)
        adj = xs.EmpiricalQuantileMapping.train(ref, hist, kind="*", group="time.month", max_tail_factor=factor)
        got = adj.adjust(sim, interp=interp).transpose("time", "lat").values
        thr = u.broadcast(adj.ds.hist_q_raw.isel(quantiles=-1), sim, group=xs.Grouper("time.month"), interp=interp)
    thr = thr.transpose("time", "lat").values.astype(np.float32)
    mask = simv.astype(np.float32) > np.float32(factor) * thr
    assert 5 < mask.sum() < 0.2 * mask.size
    near = simv > factor * adj.ds.hist_q_raw.isel(quantiles=-1).transpose("month", "lat").values[xs.date_range("2000-01-01", n, "noleap").month - 1]
    assert (near != mask).any()  # the interpolated threshold decides differently from the month's own value somewhere
    exp = np.where(mask, simv, plain.transpose("time", "lat").values)
    edge = np.isclose(simv, factor * thr, rtol=1e-6)  # float32 rounding of the threshold may flip exact ties
    np.testing.assert_array_equal(got[~edge], exp[~edge])


def _cannon_2015_rvs(n, random=True, rng=None):
    """testing/helpers.py:23-35 (Cannon et al. 2015): ref ~ gamma(4, 7.5), hist ~ gamma(8.15, 3.68), sim ~ gamma(16, 2.63)"""
    from scipy.stats import gamma

    fd = (gamma(4, scale=7.5), gamma(8.15, scale=3.68), gamma(16, scale=2.63))
    if random:
        r = [d.rvs(n, random_state=rng) for d in fd]
    else:
        uu = u.equally_spaced_nodes(n, None)
        r = [d.ppf(uu) for d in fd]
    return [timelonlatseries(x, attrs={"units": "kg/m/m/s"}) for x in r]


class TestMoreReferenceCases:
    """the remaining cases of the reference's TestDQM / TestQDM / TestQM / TestBaseAdjustment that lie on this path"""

    @pytest.mark.parametrize("kind,units", [(ADDITIVE, "K"), (MULTIPLICATIVE, "kg m-2 s-1")])
    def test_dqm_mon_u(self, kind, units, random):
        """tests/test_adjustment.py:386-432 (without add_dims): hist U, ref U + monthly cycle, a trend on sim"""
        n = 5000
        uu = random.random(n)
        x = uniform(loc=2, scale=0.1).ppf(uu)
        y = uniform(loc=4, scale=0.1).ppf(uu) + uniform(loc=0, scale=1e-7).ppf(uu)
        attrs = {"units": units, "kind": kind}
        hist, ref = timelonlatseries(x, attrs=attrs), mon_timelonlatseries(y, attrs=attrs)
        trend = np.linspace(-0.2, 0.2, n) + int(kind == MULTIPLICATIVE)
        ref_t = mon_timelonlatseries(apply_correction(y, trend, kind), attrs=attrs)
        sim = timelonlatseries(apply_correction(x, trend, kind), attrs=attrs)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")  # DQM with a non-dayofyear grouping warns like the reference (adjustment.py:608-609)
            DQM = xs.DetrendedQuantileMapping.train(ref, hist, kind=kind, group="time.month", nquantiles=5)
            p = DQM.adjust(sim)
        mqm = DQM.ds.af.values.mean(axis=-1)
        np.testing.assert_array_almost_equal(mqm, int(kind == MULTIPLICATIVE), 1)
        np.testing.assert_allclose(p.values, ref_t.values, rtol=0.1, atol=0.5)

    def test_dqm_cannon_and_from_ds(self, random):
        """tests/test_adjustment.py:434-452"""
        ref, hist, sim = _cannon_2015_rvs(15000, rng=random)
        dqm = xs.DetrendedQuantileMapping.train(ref, hist, kind="*", group="time")
        p = dqm.adjust(sim)
        np.testing.assert_almost_equal(p.values.mean(), 41.6, 0)
        np.testing.assert_almost_equal(p.values.std(), 15.0, 0)
        dqm2 = xs.DetrendedQuantileMapping.from_dataset(dqm.ds)
        for name in dqm.ds.data_vars:
            np.testing.assert_array_equal(dqm.ds[name].values, dqm2.ds[name].values)
        np.testing.assert_array_equal(p.values, dqm2.adjust(sim).values)

    def test_qdm_mon_u(self, random):
        """tests/test_adjustment.py:749-802, the multiplicative branch (interp="nearest"; the additive one asks for the
        grouped linear interpolation that goes through Qhull in the reference)"""
        kind, units = MULTIPLICATIVE, "kg m-2 s-1"
        uu = random.random(10000)
        xd, yd = uniform(loc=1, scale=1), uniform(loc=2, scale=2)
        x = xd.ppf(uu)
        y = yd.ppf(uu) + uniform(loc=0, scale=1e-7).ppf(uu)
        attrs = {"units": units, "kind": kind}
        ref = mon_timelonlatseries(y, attrs=attrs)
        hist = sim = timelonlatseries(x, attrs=attrs)
        QDM = xs.QuantileDeltaMapping.train(ref, hist, kind=kind, group="time.month", nquantiles=40)
        p = QDM.adjust(sim, interp="nearest")
        q = QDM.ds.coords["quantiles"].values
        expected = get_correction(xd.ppf(q), yd.ppf(q), kind)
        expected = apply_correction(MON_TRI[:, np.newaxis], expected[np.newaxis, :], kind)
        np.testing.assert_array_almost_equal(QDM.ds.af.values, expected, 1)
        np.testing.assert_allclose(p.values, ref.values, rtol=0.1, atol=0.2)

    def test_qdm_cannon_and_diagnostics(self):
        """tests/test_adjustment.py:836-859"""
        ref, hist, sim = _cannon_2015_rvs(15000, random=False)
        with xs.set_options(extra_output=True):
            QDM = xs.QuantileDeltaMapping.train(ref, hist, kind="*", group="time", nquantiles=50)
            scends = QDM.adjust(sim)
        assert isinstance(scends, xs.Dataset) and {"scen", "sim_q"} <= set(scends.data_vars)
        np.testing.assert_almost_equal(scends["scen"].values.mean(), 41.5, 1)
        np.testing.assert_almost_equal(scends["scen"].values.std(), 16.7, 0)

    def test_eqm_different_times_training(self, random):
        """tests/test_adjustment.py:1072-1083"""
        n = 10
        uu, u2 = random.random(n), random.random(n)
        ref = timelonlatseries(uu, start="2000-01-01", attrs={"units": "K"})
        hist = timelonlatseries(u2, start="2000-01-01", attrs={"units": "K"})
        hist_fut = timelonlatseries(u2, start="2001-01-01", attrs={"units": "K"})
        ds = xs.EmpiricalQuantileMapping.train(ref, hist).ds
        with pytest.raises(ValueError):
            xs.EmpiricalQuantileMapping.train(ref, hist_fut)
        xs.EmpiricalQuantileMapping._allow_diff_training_times = True
        try:
            ds_fut = xs.EmpiricalQuantileMapping.train(ref, hist_fut).ds
        finally:
            xs.EmpiricalQuantileMapping._allow_diff_training_times = False
        np.testing.assert_array_equal(ds.af.values, ds_fut.af.values)

    def test_base_matching_times_and_units(self, random):
        """tests/test_adjustment.py:61-116: units are required and must agree; ref / hist time axes must match"""
        n = 10
        uu = random.random(n)
        da = timelonlatseries(uu, attrs={"units": "K"})
        da2 = timelonlatseries(uu, attrs={"units": "degC"})
        # different but compatible units are harmonized (adjustment.py:182-185): hist (degC) comes to the units of ref (K)
        eq = xs.EmpiricalQuantileMapping.train(da, da2, group="time", nquantiles=3)
        np.testing.assert_allclose(eq.ds.af.values, -273.15, atol=1e-3)
        with pytest.raises(ValueError, match="different dimensions"):
            xs.EmpiricalQuantileMapping.train(da, timelonlatseries(uu, attrs={"units": "mm d-1"}))
        with pytest.raises(ValueError, match="units"):
            xs.EmpiricalQuantileMapping.train(da, timelonlatseries(uu))
        short = timelonlatseries(uu[:-1], attrs={"units": "K"})
        with pytest.raises(ValueError):
            xs.EmpiricalQuantileMapping.train(da, short)


@pytest.mark.parametrize("interp,expi", [("nearest", 2.9), ("linear", 2.95)])
@pytest.mark.parametrize("extrap,expe", [("constant", 4.4), ("nan", np.nan)])
def test_interp_on_quantiles_constant(interp, expi, extrap, expe):
    # reference tests/test_utils.py:68-113 ('cubic' is outside the device path)
    quantiles = np.linspace(0, 1, num=25)
    xq = DataArray(np.linspace(205, 229, num=25), ("quantiles",), {"quantiles": quantiles})
    yq = DataArray(np.linspace(2, 4.4, num=25), ("quantiles",), {"quantiles": quantiles})
    v = np.linspace(240, 200, num=41) - 0.5
    newx = DataArray(np.where(v > 201, v, np.nan), ("time",), {"time": xs.date_range("1900-03-01", 41, "standard").values})
    xq, yq, newx = xq.expand_dims(lat=[1, 2, 3]), yq.expand_dims(lat=[1, 2, 3]), newx.expand_dims(lat=[1, 2, 3])
    for punch in (False, True):
        if punch:
            xq, yq = xq.where(xq != 220), yq.where(yq != 3)
        out = u.interp_on_quantiles(newx, xq, yq, group="time", method=interp, extrapolation=extrap)
        if np.isnan(expe):
            assert out.isel(time=0).isnull().all()
        else:
            assert out.isel(lat=1, time=0) == expe
        np.testing.assert_allclose(out.isel(time=25).values, expi)
        assert out.isel(time=-1).isnull().all()


def test_interp_on_quantiles_monthly(random):
    # reference tests/test_utils.py:116-146: no NaN with constant extrapolation on the month path
    T = 365 * 31
    t = xs.date_range("2000-01-01", T, "noleap")
    doy, days = t.dayofyear, np.arange(T)
    mk = lambda v: DataArray(v, ("time",), {"time": t.values}, {"units": "K"})  # noqa: E731
    ref = mk(-20 * np.cos(2 * np.pi * doy / 365) + 2 * random.random(T) + 273.15 + 0.1 * days / 365)
    sim = mk(-18 * np.cos(2 * np.pi * doy / 365) + 2 * random.random(T) + 273.15 + 0.11 * days / 365)
    n = 365 * 15
    refc, hist = ref.isel(time=slice(0, n)), sim.isel(time=slice(0, n))
    group = xs.Grouper("time.month")
    quantiles = u.equally_spaced_nodes(15, eps=1e-6)
    ref_q = group.apply(nbu.quantile, refc, main_only=True, q=quantiles)
    hist_q = group.apply(nbu.quantile, hist, main_only=True, q=quantiles)
    af = u.get_correction(hist_q, ref_q, "+")
    # the reference loops over all three methods (:144-146); "linear" and "cubic" go through the per-cell Delaunay
    # triangulation here (piecewise-linear / Clough-Tocher interpolant, like scipy.griddata)
    for interp in ["nearest", "linear", "cubic"]:
        afi = u.interp_on_quantiles(sim, hist_q, af, group="time.month", method=interp, extrapolation="constant")
        assert afi.isnull().sum("time") == 0, interp


def test_rank(random):
    # reference tests/test_utils.py:197-205
    arr = random.random((10, 10, 1000))
    ranks = u.rank(DataArray(arr, ("x", "y", "time")), dim="time", pct=False)
    np.testing.assert_array_equal(ranks.values, arr.argsort().argsort() + 1)
    # ties get average ranks (:208-217 data)
    tie = u.rank(DataArray(np.array([1, 26, 2, 4.0, 6, 2, 2]), ("time",)), dim="time", pct=False)
    np.testing.assert_array_equal(tie.values, [1, 7, 3, 5, 6, 3, 3])


def test_nbutils_quantile(random):
    # reference tests/test_nbutils.py: equals nanquantile to 6 decimals; single-value and all-NaN edge cases
    x = random.gamma(2.0, 3.0, (5, 2190))
    da = DataArray(x, ("location", "time"))
    q = np.linspace(0.1, 0.99, 50)
    out = nbu.quantile(da, q, dim="time")
    assert out.dims == ("location", "quantiles")
    np.testing.assert_array_almost_equal(out.values, np.quantile(x, q, axis=1).T)
    one = DataArray(np.array([1] + [np.nan] * 100), ("dim_0",))
    np.testing.assert_array_equal(nbu.quantile(one, q, dim="dim_0").values, np.full_like(q, 1))
    allnan = DataArray(np.array([np.nan] * 100), ("dim_0",))
    np.testing.assert_array_equal(nbu.quantile(allnan, q, dim="dim_0").values, np.full_like(q, np.nan))


def test_poly_detrend():
    # reference tests/test_detrending.py:18-30
    x = timeseries(np.arange(20 * 365.25), units="K")
    fx = xs.PolyDetrend(degree=1).fit(x)
    dx = fx.detrend(x)
    xt = fx.retrend(dx)
    np.testing.assert_array_almost_equal(dx.values, 0)
    np.testing.assert_array_almost_equal(xt.values, x.values)
    fx2 = xs.PolyDetrend.from_dataset(fx.ds)
    np.testing.assert_array_equal(fx2.detrend(x).values, dx.values)


class TestPreprocessingInsideTheClasses:
    """reference tests/test_adjustment.py:494-499 (adapt_freq grouping), :585-596 (explicit adapt_freq == the option),
    :1085-1130 (jitter inside == jitter outside, on synthetic data instead of the gosset files), :61-70 (harmonize_units)"""

    @pytest.mark.parametrize("group", ["time", "time.month"])
    def test_adapt_freq_grouping(self, group, random):
        ref, hist, sim = _cannon_2015_rvs(15000, rng=random)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            dqm = xs.DetrendedQuantileMapping.train(ref, hist, kind="*", group=group, adapt_freq_thresh="1 kg m-2 d-1")
        assert np.isfinite(dqm.adjust(sim).values).all()
        for v in ("P0_ref", "P0_hist", "pth"):
            assert v in dqm.ds.data_vars

    def test_adapt_freq_time_explicit(self, random):
        from xsdba_amd.processing import adapt_freq, jitter_under_thresh

        ref, hist, _ = _cannon_2015_rvs(15000, rng=random)
        thr = "1 kg m-2/d"
        np.random.seed(3)
        ref = jitter_under_thresh(ref, "0.1   kg m-2 / d")
        hist = jitter_under_thresh(hist, "0.1 kg m-2 / d")
        np.random.seed(5)
        hist_ad, _, _ = adapt_freq(ref, hist, group="time", thresh=thr)
        np.random.seed(5)  # the same draw inside the class (one integer of numpy's generator seeds the device stream)
        adj = xs.DetrendedQuantileMapping.train(ref, hist, kind="*", group="time", adapt_freq_thresh=thr)
        np.random.seed(7)
        out = adj.adjust(hist)
        adj.adapt_freq_thresh = None
        out_ad = adj.adjust(hist_ad)
        # the values the adaptation replaced are random draws (other seed at adjust time): compare where nothing was replaced
        same = hist.values == hist_ad.values
        assert same.mean() > 0.9
        np.testing.assert_allclose(out.values[same], out_ad.values[same], rtol=1e-5)

    def test_jitter_under_and_over_thresh_inside(self, random):
        from xsdba_amd.processing import jitter_over_thresh, jitter_under_thresh

        n = 12000
        pr = lambda s: timeseries(np.where(random.random(n) < 0.3, 0.0, random.gamma(2.0, s, n)), units="mm/d")  # noqa: E731
        ref, hist = pr(3.0), pr(4.0)
        np.random.seed(42)
        inside = xs.EmpiricalQuantileMapping.train(ref, hist, jitter_under_thresh_value="0.01 mm/d", group="time").ds.af.values
        np.random.seed(42)
        outside = xs.EmpiricalQuantileMapping.train(ref, jitter_under_thresh(hist, "0.01 mm/d"), group="time").ds.af.values
        np.testing.assert_array_equal(inside, outside)  # the same seed gives the same device stream
        tas = lambda m: timeseries(np.clip(m + random.standard_normal(n), None, 2.5), units="K")  # noqa: E731
        ref, hist = tas(1.0), tas(1.5)
        np.random.seed(42)
        inside = xs.EmpiricalQuantileMapping.train(ref, hist, jitter_over_thresh_value="2 K", jitter_over_thresh_upper_bnd="3 K", group="time").ds.af.values
        np.random.seed(42)
        outside = xs.EmpiricalQuantileMapping.train(ref, jitter_over_thresh(hist, "2 K", "3 K"), group="time").ds.af.values
        np.testing.assert_array_equal(inside, outside)

    def test_harmonize_units(self, random):
        from xsdba_amd.adjustment import BaseAdjustment
        from xsdba_amd.units import convert_units_to

        da = timelonlatseries(random.random(10) + 280, attrs={"units": "K", "kind": ADDITIVE})
        da2 = convert_units_to(da, "degC")
        assert da2.attrs["units"] == "degC"
        np.testing.assert_allclose(da2.values, da.values - 273.15, atol=1e-4)
        (a, b), _ = BaseAdjustment._harmonize_units(da, da2)
        assert a.attrs["units"] == b.attrs["units"] == "K"
        np.testing.assert_allclose(b.values, da.values, atol=1e-4)
