"""The reference's tests of the pre-processing steps in front of the quantile kernels, re-expressed
against xsdba_amd (GPU): tests/test_processing.py :31-64 (jitter), :66-88 (narrow intervals), :91-132 (adapt_freq),
tests/test_adjustment.py :586-596 (adapt_freq inside DQM == outside), :1083-1128 (jitter inside EQM == outside)."""
from __future__ import annotations

import warnings

import numpy as np
import pytest

import oracle
import xsdba_amd as xs
from xsdba_amd.labeled import DataArray
from xsdba_amd.processing import adapt_freq, jitter, jitter_over_thresh, jitter_under_thresh

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    from xsdba_amd import hip

    hip.default_context(0)


def _da(vals, units, start="2000-01-01", calendar="standard", dims=None, extra=None):
    vals = np.asarray(vals)
    dims = dims or ("time",) + (("lat",) if vals.ndim == 2 else ())
    coords = {"time": xs.date_range(start, vals.shape[0], calendar).values}
    if vals.ndim == 2:
        coords["lat"] = np.arange(vals.shape[1])
    return DataArray(vals, dims, coords, {"units": units, **(extra or {})})


def test_jitter_both():
    da = _da([0.5, 2.1, np.nan], "K")
    out = jitter(da, lower="1 K", upper="2 K", maximum="3 K")
    assert da.values[0] != out.values[0] and 0 < out.values[0] < 1
    assert da.values[1] != out.values[1] and 2 <= out.values[1] < 3
    assert np.isnan(out.values[2])


def test_jitter_under_thresh():
    da = _da([0.5, 2.1, np.nan], "K")
    out = jitter_under_thresh(da, "1 K")
    assert da.values[0] != out.values[0] and 0 < out.values[0] < 1
    np.testing.assert_allclose(da.values[1:], out.values[1:])
    assert out.attrs["units"] == "K"


def test_jitter_over_thresh():
    da = _da([0.5, 2.1, np.nan], "m")
    out = jitter_over_thresh(da, "2 m", "3 m")
    assert da.values[1] != out.values[1] and 2 <= out.values[1] < 3
    np.testing.assert_allclose(da.values[[0, 2]], out.values[[0, 2]])
    out = jitter_over_thresh(da, "200 cm", "0.003 km")  # thresholds in other units of the same dimension (test_processing.py:57-63)
    assert da.values[1] != out.values[1] and 2 <= out.values[1] < 3
    with pytest.raises(ValueError, match="cannot be expressed"):
        jitter_over_thresh(da, "2 K", "3 K")


@pytest.mark.parametrize("test_val", [1e-6, 1, 100, 1e6])
@pytest.mark.parametrize("dtype, delta", [("f8", 1e-8), ("f4", 1e-6)])
def test_jitter_other_dtypes(dtype, delta, test_val):
    # narrow intervals are meant to increase the likelihood of rounding issues (tests/test_processing.py:66-75)
    da = _da(test_val + np.zeros(1000, dtype=dtype), "%")
    out_high = jitter(da, upper=f"{test_val * (1 - delta):.20f} %", maximum=f"{test_val:.20f} %")
    out_low = jitter(da, lower=f"{test_val * (1 + delta):.20f} %", minimum=f"{test_val:.20f} %")
    assert (out_high.values < test_val).all()
    assert (out_low.values > test_val).all()
    assert out_high.dtype == np.dtype(dtype)


def test_adapt_freq():
    """tests/test_processing.py:91-132"""
    rng = np.random.default_rng(0)
    n = 31 * 365 + 8  # 1990-01-01 .. 2020-12-31
    pr = rng.integers(0, 100, size=(n, 3)).astype(np.float64)
    prsim = _da(np.where(pr < 20, pr / 20, pr), "mm d-1", start="1990-01-01")
    prref = _da(np.where(pr < 10, pr / 20, pr), "mm d-1", start="1990-01-01")
    sim_ad, pth, dP0 = adapt_freq(prref, prsim, thresh="1 mm d-1", group=xs.Grouper("time.month"))
    input_zeros = np.where(prsim.values <= 1, sim_ad.transpose("time", "lat").values, np.nan)
    # the proportion of corrected values (n * 3 * 0.2 is the theoretical number of values under 1 in prsim)
    dP0_out = np.nansum(input_zeros > 1) / (n * 3 * 0.2)
    np.testing.assert_allclose(dP0_out, 0.5, atol=0.1)
    corrected = input_zeros[input_zeros > 1]
    assert ((corrected < 20.1) & (corrected > 1)).all()
    keep = prsim.values > 20.5
    np.testing.assert_array_equal(sim_ad.transpose("time", "lat").values[keep], prsim.values[keep])
    np.testing.assert_allclose(pth.values, 20, rtol=0.05)
    np.testing.assert_allclose(dP0.values, 0.5, atol=0.25)
    assert sim_ad.attrs["units"] == "mm d-1" and sim_ad.attrs["references"].startswith("Themeßl")
    assert pth.attrs["units"] == "mm d-1" and pth.dims == ("lat", "month")


def _cannon(n, rng, units="kg m-2 d-1"):
    """testing/helpers.py:23-35 (Cannon et al. 2015 gamma variates)"""
    ref = _da(rng.gamma(4.0, 7.5, n), units)
    hist = _da(rng.gamma(8.15, 3.68, n), units)
    sim = _da(rng.gamma(16.0, 2.63, n), units, start="2050-01-01")
    return ref, hist, sim


def test_adapt_freq_time_explicit():
    """tests/test_adjustment.py:586-596: adapt_freq inside DQM == adapt_freq outside, then DQM without it.  numpy's
    global generator is re-seeded before each call so that both branches take the same draws."""
    rng = np.random.default_rng(3)
    ref, hist, _ = _cannon(15000, rng)
    thr = "1 kg m-2 d-1"
    ref, hist = jitter_under_thresh(ref, "0.1 kg m-2 d-1"), jitter_under_thresh(hist, "0.1 kg m-2 d-1")
    # make the frequency adaptation matter: more dry days in hist than in ref
    hv = hist.values.copy()
    hv[::3] = 0.05
    hist = DataArray(hv, hist.dims, hist.coords, hist.attrs)
    np.random.seed(11)
    hist_ad, pth, dP0 = adapt_freq(ref, hist, group="time", thresh=thr)
    assert float(dP0.values.ravel()[0]) > 0.2 and (hist_ad.values != hist.values).sum() > 1000
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        np.random.seed(11)
        ADJ = xs.DetrendedQuantileMapping.train(ref, hist, kind="*", group="time", adapt_freq_thresh=thr)
        assert {"P0_ref", "P0_hist", "pth"} <= set(ADJ.ds.data_vars)
        np.testing.assert_allclose(ADJ.ds.pth.values.ravel(), pth.values.ravel())
        np.random.seed(11)
        out = ADJ.adjust(hist)
        ADJ.adapt_freq_thresh = None
        out_ad = ADJ.adjust(hist_ad)
    np.testing.assert_allclose(out.values, out_ad.values)


def test_jitter_inside_equals_outside():
    """tests/test_adjustment.py:1083-1128"""
    rng = np.random.default_rng(5)
    ref, hist, _ = _cannon(9000, rng)
    np.random.seed(42)
    af_in = xs.EmpiricalQuantileMapping.train(ref, hist, jitter_under_thresh_value="10 kg m-2 d-1", group="time").ds.af
    np.random.seed(42)
    af_out = xs.EmpiricalQuantileMapping.train(ref, jitter_under_thresh(hist, "10 kg m-2 d-1"), group="time").ds.af
    np.testing.assert_array_equal(af_in.values, af_out.values)
    np.random.seed(42)
    af_in = xs.EmpiricalQuantileMapping.train(ref, hist, jitter_over_thresh_value="40 kg m-2 d-1",
                                              jitter_over_thresh_upper_bnd="50 kg m-2 d-1", group="time").ds.af
    np.random.seed(42)
    af_out = xs.EmpiricalQuantileMapping.train(ref, jitter_over_thresh(hist, "40 kg m-2 d-1", "50 kg m-2 d-1"), group="time").ds.af
    np.testing.assert_array_equal(af_in.values, af_out.values)
    with pytest.raises(ValueError) as err:  # re-wrapped by map_blocks like in the reference (base.py:701-702)
        xs.EmpiricalQuantileMapping.train(ref, hist, jitter_over_thresh_value="40 kg m-2 d-1", group="time")
    assert "must both be specified" in str(err.value.__cause__)


@pytest.mark.parametrize("cls", ["EmpiricalQuantileMapping", "DetrendedQuantileMapping", "QuantileDeltaMapping"])
def test_adapt_freq_doy_window(cls):
    """adapt_freq + jitter with windowed day-of-year groups (the blocks pool [time, window]); the trained dataset
    carries P0_ref / P0_hist / pth and the adjust step re-uses them on the groups without window"""
    rng = np.random.default_rng(8)
    n = 6 * 365
    pr = rng.uniform(0, 40, (n, 2))
    mk = lambda v, start: DataArray(v, ("time", "lat"), {"time": xs.date_range(start, n, "noleap").values, "lat": np.arange(2)},  # noqa: E731
                                    {"units": "mm d-1"})
    ref = mk(np.where(pr < 10, 0.0, pr).astype(np.float32), "2000-01-01")
    hist = mk(np.where(pr < 20, 0.0, pr).astype(np.float32), "2000-01-01")
    sim = mk(np.where(pr[::-1] < 22, 0.0, pr[::-1]).astype(np.float32), "2040-01-01")
    group = xs.Grouper("time.dayofyear", window=31)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        adj = getattr(xs, cls).train(ref, hist, kind="*", group=group, nquantiles=15, adapt_freq_thresh="1 mm d-1",
                                     jitter_under_thresh_value="0.01 mm d-1")
        assert adj.ds.pth.dims == ("lat", "dayofyear") and adj.ds.P0_hist.shape == (2, 365)
        np.testing.assert_allclose(adj.ds.P0_hist.values, 0.5, atol=0.17)
        np.testing.assert_allclose(adj.ds.P0_ref.values, 0.25, atol=0.15)
        scen = adj.adjust(sim)
    v = scen.transpose("lat", "time").values
    assert np.isfinite(v).all() and scen.dtype == np.float32
    # about half of sim's dry days were given a wet value before the mapping: the dry-day frequency moved towards ref's
    assert 0.2 < (v < 1).mean() < 0.42
    assert "adapt_freq_thresh='1 mm d-1'" in adj.ds.attrs["adj_params"]


@pytest.mark.parametrize("gname,window", [("time.month", 1), ("time.dayofyear", 31)])
def test_adapt_freq_add_dims_are_not_pooled(gname, window):
    """Grouper(add_dims=["member"]) in adapt_freq (reference _processing.py:85-86: "ADD_DIMS are not pooled with other dims for
    frequency adaptation"; tests/test_processing.py:166-182: the extra dimension stays on pth): every member is adapted on its own
    — pth and dP0 equal those of the same call without `add_dims`, where the member is one more cell dimension — and a ref that
    lacks the dimension is broadcast along it (_processing.py:104-106)."""
    rng = np.random.default_rng(3)
    T, L, M = 4 * 365, 2, 3
    pr = rng.integers(0, 100, size=(T, L, M)).astype(np.float64)
    sim_v, ref_v = np.where(pr < 20, pr / 20, pr), np.where(pr < 10, pr / 20, pr)
    crd = {"time": xs.date_range("2001-01-01", T, "noleap").values, "lat": np.arange(L), "member": np.arange(M)}
    sim = DataArray(sim_v, ("time", "lat", "member"), crd, {"units": "mm d-1"})
    ref = DataArray(ref_v, ("time", "lat", "member"), crd, {"units": "mm d-1"})
    gr = xs.Grouper(gname, window=window, add_dims=["member"])
    sim_ad, pth, dP0 = adapt_freq(ref, sim, thresh="1 mm d-1", group=gr)
    assert set(sim_ad.dims) == set(sim.dims) and sim_ad.transpose(*sim.dims).shape == sim.shape
    assert "member" in pth.dims and "member" in dP0.dims
    plain = xs.Grouper(gname, window=window)
    _, pth0, dP00 = adapt_freq(ref, sim, thresh="1 mm d-1", group=plain)
    order = ("lat", "member", gr.prop)
    np.testing.assert_array_equal(dP0.transpose(*order).values, dP00.transpose(*order).values)
    np.testing.assert_allclose(pth.transpose(*order).values, pth0.transpose(*order).values, rtol=1e-12, equal_nan=True)
    # against the oracle member by member: P0 from that member's windows alone
    gidx, G = plain.group_index(sim)
    for lat in range(L):
        for m in range(M):
            _, epth, edp0, _, _ = oracle.adapt_freq(sim_v[:, lat, m][None], ref_v[:, lat, m][None], np.asarray(gidx, np.int32), G, window, 1.0, seed=0)
            np.testing.assert_allclose(dP0.transpose(*order).values[lat, m], edp0.ravel(), rtol=1e-12, equal_nan=True)
            np.testing.assert_allclose(pth.transpose(*order).values[lat, m], epth.ravel(), rtol=1e-12, equal_nan=True)
    v = sim_ad.transpose(*sim.dims).values
    changed = v != sim_v
    assert changed.any() and (sim_v[changed] <= 1).all() and (v[changed] > 1).all()
    # a ref without the extra dimension: broadcast along it (every member against the same ref)
    r0 = ref.isel(member=0)
    sim_b, pth_b, dP0_b = adapt_freq(r0, sim, thresh="1 mm d-1", group=gr)
    assert "member" in pth_b.dims and set(sim_b.dims) == set(sim.dims)
    for m in range(M):
        _, _, edp0, _, _ = oracle.adapt_freq(sim_v[:, 0, m][None], ref_v[:, 0, 0][None], np.asarray(gidx, np.int32), G, window, 1.0, seed=0)
        np.testing.assert_allclose(dP0_b.transpose(*order).values[0, m], edp0.ravel(), rtol=1e-12, equal_nan=True)


def test_adapt_freq_sixty_years_one_group():
    """the size the fuzzer met round 4's row limits at (60 years of daily float64 data under group="time": a block of 21 900
    values, beyond the LDS sorts): through the public functions, inside a training and outside"""
    rng = np.random.default_rng(21)
    n = 60 * 365
    ref = _da(np.where(rng.random(n) < 0.3, 0.0, rng.gamma(0.8, 6.0, n)), "mm d-1", start="1950-01-01")
    hist = _da(np.where(rng.random(n) < 0.6, 0.0, rng.gamma(0.8, 6.0, n)), "mm d-1", start="1950-01-01")
    hist_ad, pth, dP0 = adapt_freq(ref, hist, group="time", thresh="1 mm d-1")
    d = float(dP0.values.ravel()[0])
    assert 0.2 < d < 0.6
    changed = hist_ad.values != hist.values
    # the share of the sub-threshold values that were lifted is dP0 (the band of ranks between P0_ref / P0_hist * P0_sim and P0_sim)
    sub = hist.values <= 1.0
    assert not changed[~sub].any()
    np.testing.assert_allclose(changed[sub].mean(), d, atol=2e-3)
    lifted = hist_ad.values[changed]
    assert (lifted > 1.0).all() and (lifted <= float(pth.values.ravel()[0]) + 1e-12).all()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ADJ = xs.EmpiricalQuantileMapping.train(ref, hist, kind="*", group="time", adapt_freq_thresh="1 mm d-1")
    np.testing.assert_allclose(ADJ.ds.pth.values.ravel(), pth.values.ravel())
