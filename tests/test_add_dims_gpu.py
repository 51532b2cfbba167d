"""Grouper(add_dims=...) (reference base.py:413-416: the extra dimensions are pooled into every group's sample):
xsdba_amd lays the pooled members end to end on the time axis (xsdba_amd/_adjustment.py).  The expectations here do
NOT use that layout: every group's pooled, windowed row is gathered explicitly (rolling.construct semantics,
base.py:261-262) and handed to the oracle as a one-group series.  Mirrors the reference's tests
tests/test_adjustment.py:119-151 (error), :258-276 (Scaling), :281-316 and :393-433 (DQM), :748-800 (QDM)."""
from __future__ import annotations

import numpy as np
import pytest
from scipy.stats import rankdata, uniform

import oracle
import xsdba_amd as xs
from xsdba_amd.labeled import DataArray
from xsdba_amd.utils import apply_correction

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    from xsdba_amd import hip

    hip.default_context(0)


def cube(values, dims, units="K", calendar="noleap", start="2000-01-01"):
    values = np.asarray(values)
    coords = {"time": xs.date_range(start, values.shape[dims.index("time")], calendar).values}
    for d, n in zip(dims, values.shape):
        if d != "time":
            coords[d] = np.arange(n)
    return DataArray(values, dims, coords, {"units": units})


def pooled_rows(x3, gidx, G, W):
    """x3 [member, time] -> per group the pooled windowed sample (every centre of the group, every member, every
    window slot; slots outside the series are NaN)"""
    P, T = x3.shape
    h = W // 2
    rows = []
    for g in range(G):
        tc = np.nonzero(gidx == g)[0]
        vals = []
        for d in range(-h, h + 1):
            t = tc + d
            ok = (t >= 0) & (t < T)
            vals.append(x3[:, t[ok]].ravel())
            vals.append(np.full(P * int((~ok).sum()), np.nan, x3.dtype))
        rows.append(np.concatenate(vals))
    return rows


def Z(row):
    return np.zeros(row.shape[-1], np.int32)


def series(rng, shape, dtype=np.float32, nan_frac=0.0):
    x = (280 + 10 * rng.standard_normal(shape)).astype(dtype)
    if nan_frac:
        x[rng.random(shape) < nan_frac] = np.nan
    return x


GROUPS = [("time", 1), ("time.month", 1), ("time.dayofyear", 31), ("time.dayofyear", 1)]


@pytest.mark.parametrize("group,window", GROUPS)
@pytest.mark.parametrize("kind", ["+", "*"])
@pytest.mark.parametrize("method", ["eqm", "dqm"])
def test_train_pools_members(group, window, kind, method):
    rng = np.random.default_rng(5)
    L, P, T, Q = 2, 3, 4 * 365, 12
    ref = series(rng, (L, P, T), nan_frac=0.02)
    hist = series(rng, (L, P, T), nan_frac=0.02) + 2
    gr = xs.Grouper(group, window, add_dims=["member"])
    cls = xs.EmpiricalQuantileMapping if method == "eqm" else xs.DetrendedQuantileMapping
    obj = cls.train(cube(ref, ("lat", "member", "time")), cube(hist, ("lat", "member", "time")), group=gr, kind=kind, nquantiles=Q)
    assert "member" not in obj.ds.af.dims
    gidx, G = gr.group_index(cube(ref, ("lat", "member", "time")))
    q = np.asarray(obj.ds.quantiles.values, np.float32)
    prop = gr.prop
    af = obj.ds.af.transpose("lat", prop, "quantiles").values
    hq = obj.ds.hist_q.transpose("lat", prop, "quantiles").values
    for c in range(L):
        rr, hh = pooled_rows(ref[c], gidx, G, window), pooled_rows(hist[c], gidx, G, window)
        for g in (range(G) if G <= 12 else list(range(0, G, 37)) + [G - 1]):
            if method == "eqm":
                eaf, ehq = oracle.eqm_train(rr[g][None], hh[g][None], Z(hh[g]), 1, 1, q, kind)
            else:
                eaf, ehq, esc = oracle.dqm_train(rr[g][None], hh[g][None], Z(hh[g]), 1, 1, q, kind)
                sc = obj.ds.scaling.transpose("lat", prop).values
                np.testing.assert_allclose(sc[c, g], esc[0, 0], rtol=2e-6)
            tol = dict(rtol=0, atol=0) if method == "eqm" else dict(rtol=3e-6, atol=3e-5)
            np.testing.assert_allclose(hq[c, g], ehq[0, 0], **tol)
            np.testing.assert_allclose(af[c, g], eaf[0, 0], rtol=1e-5 if method == "dqm" else 0, atol=1e-4 if method == "dqm" else 0)


@pytest.mark.parametrize("which", ["ref", "hist"])
@pytest.mark.parametrize("method", ["eqm", "dqm", "scaling"])
def test_pooled_dim_in_only_one_dataset(which, method):
    """reference tests/test_adjustment.py:552-582 and :258-276: `add_dims` present in ref or hist only"""
    rng = np.random.default_rng(6)
    P, T, Q = 3, 3 * 365, 10
    ref = series(rng, (P, T)) if which == "ref" else series(rng, (T,))
    hist = series(rng, (P, T)) + 1 if which == "hist" else series(rng, (T,)) + 1
    dims = lambda a: ("member", "time") if a.ndim == 2 else ("time",)  # noqa: E731
    gr = xs.Grouper("time.month", add_dims=["member"])
    r, h = cube(ref, dims(ref)), cube(hist, dims(hist))
    gidx, G = gr.group_index(r)
    rr = pooled_rows(np.atleast_2d(ref), gidx, G, 1)
    hh = pooled_rows(np.atleast_2d(hist), gidx, G, 1)
    if method == "scaling":
        obj = xs.Scaling.train(r, h, group=gr, kind="+")
        assert "member" not in obj.ds.af.dims
        exp = np.array([np.nanmean(rr[g].astype(np.float64)) - np.nanmean(hh[g].astype(np.float64)) for g in range(G)])
        np.testing.assert_allclose(obj.ds.af.values.ravel(), exp, atol=1e-4)  # fp32 means of values near 280
        p = obj.adjust(h)
        assert p.values.shape == hist.shape
        return
    cls = xs.EmpiricalQuantileMapping if method == "eqm" else xs.DetrendedQuantileMapping
    obj = cls.train(r, h, group=gr, kind="+", nquantiles=Q)
    q = np.asarray(obj.ds.quantiles.values, np.float32)
    for g in range(G):
        if method == "eqm":
            ehq = oracle.quantile_groups(hh[g][None], Z(hh[g]), 1, 1, q)[0, 0]
            erq = oracle.quantile_groups(rr[g][None], Z(rr[g]), 1, 1, q)[0, 0]
            np.testing.assert_array_equal(obj.ds.hist_q.values[g], ehq)
            np.testing.assert_array_equal(obj.ds.af.values[g], erq - ehq)
        else:
            _, ehq, _ = oracle.dqm_train(hh[g][None], hh[g][None], Z(hh[g]), 1, 1, q, "+")
            _, erq, _ = oracle.dqm_train(rr[g][None], rr[g][None], Z(rr[g]), 1, 1, q, "+")
            np.testing.assert_allclose(obj.ds.hist_q.values[g], ehq[0, 0], atol=3e-5)
            np.testing.assert_allclose(obj.ds.af.values[g], erq[0, 0] - ehq[0, 0], atol=6e-5)
    p = obj.adjust(h)
    assert np.isfinite(p.values).all() and p.values.shape == hist.shape


@pytest.mark.parametrize("group,window", [("time.month", 1), ("time.dayofyear", 31)])
def test_eqm_adjust_broadcasts_the_tables_over_members(group, window):
    """adjust keeps the pooled dims (map_blocks reduces only <prop>, quantiles: _adjustment.py:594): every member is mapped
    through the table of its cell"""
    rng = np.random.default_rng(7)
    L, P, T, Q = 2, 3, 3 * 365, 15
    ref, hist = series(rng, (L, P, T)), series(rng, (L, P, T)) + 2
    sim = series(rng, (L, P, T), nan_frac=0.01) + 3
    gr = xs.Grouper(group, window, add_dims=["member"])
    dims = ("lat", "member", "time")
    eqm = xs.EmpiricalQuantileMapping.train(cube(ref, dims), cube(hist, dims), group=gr, nquantiles=Q)
    p = eqm.adjust(cube(sim, dims)).transpose(*dims).values
    gidx, G = gr.group_index(cube(sim, dims))
    prop = gr.prop
    af = np.ascontiguousarray(eqm.ds.af.transpose("lat", prop, "quantiles").values)
    hq = np.ascontiguousarray(eqm.ds.hist_q.transpose("lat", prop, "quantiles").values)
    for m in range(P):
        exp = oracle.qm_adjust(np.ascontiguousarray(sim[:, m]), af, hq, gidx, "nearest", "constant", "+")
        np.testing.assert_array_equal(p[:, m], exp)


@pytest.mark.parametrize("rank_window", [False, True])
def test_qdm_ranks_with_pooled_members(rank_window):
    """_adjustment.py:873: main_only = not rank_window — the default ranks every member on its own along time; with
    rank_window the ranks are taken over (time, window, pooled members)"""
    rng = np.random.default_rng(8)
    P, T, Q, W = 3, 3 * 365, 10, 31
    ref, hist = series(rng, (P, T), np.float64), series(rng, (P, T), np.float64) + 2
    sim = np.round(series(rng, (P, T), np.float64), 1)  # ties
    gr = xs.Grouper("time.dayofyear", W, add_dims=["member"])
    dims = ("member", "time")
    qdm = xs.QuantileDeltaMapping.train(cube(ref, dims), cube(hist, dims), group=gr, nquantiles=Q)
    with xs.set_options(extra_output=True):
        out = qdm.adjust(cube(sim, dims), rank_window=rank_window)
    sim_q = out.sim_q.transpose(*dims).values
    gidx, G = gr.group_index(cube(sim, dims))
    h = W // 2

    def pct(row, v):
        r = rankdata(row, "average")
        pr = r / len(row)
        mn, mx = pr.min(), pr.max()
        less, eq = (row < v).sum(), (row == v).sum()
        return mx * ((less + (eq + 1) / 2) / len(row) - mn) / (mx - mn)

    for t in list(range(0, T, 97)) + [0, T - 1]:
        g = gidx[t]
        tc = np.nonzero(gidx == g)[0]
        for m in range(P):
            if rank_window:
                tt = np.unique(np.concatenate([tc + d for d in range(-h, h + 1)]))
                tt = tt[(tt >= 0) & (tt < T)]
                row = sim[:, tt].ravel()
            else:
                row = sim[m, tc]
            np.testing.assert_allclose(sim_q[m, t], pct(row, sim[m, t]), rtol=1e-12, atol=1e-15)
    scen = out.scen.transpose(*dims).values
    assert np.isfinite(scen).all()


@pytest.mark.parametrize("group,window", [("time.month", 1), ("time.dayofyear", 31)])
@pytest.mark.parametrize("kind", ["+", "*"])
def test_dqm_adjust_fits_the_trend_on_the_pooled_mean(group, window, kind):
    """detrending.py:196-203: with pooled dims the polynomial is fitted on the mean over them (and the window)"""
    rng = np.random.default_rng(9)
    P, T, Q = 3, 4 * 365, 12
    ref, hist = series(rng, (P, T)), series(rng, (P, T)) + 2
    sim = (series(rng, (P, T)) + 3 + np.linspace(0, 4, T)[None, :]).astype(np.float32)
    gr = xs.Grouper(group, window, add_dims=["member"])
    dims = ("member", "time")
    dqm = xs.DetrendedQuantileMapping.train(cube(ref, dims), cube(hist, dims), group=gr, kind=kind, nquantiles=Q)
    from xsdba_amd._adjustment import dqm_adjust

    out = dqm_adjust(dqm.ds.assign(sim=cube(sim, dims)), group=gr, interp="nearest", kind=kind, extrapolation="constant", detrend=1)
    np.testing.assert_array_equal(dqm.adjust(cube(sim, dims), detrend=1).transpose(*dims).values, out.scen.transpose(*dims).values)
    gidx, G = gr.group_index(cube(sim, dims))
    prop = gr.prop
    af = np.ascontiguousarray(dqm.ds.af.transpose(prop, "quantiles").values)[None]
    hq = np.ascontiguousarray(dqm.ds.hist_q.transpose(prop, "quantiles").values)[None]
    sc = np.ascontiguousarray(dqm.ds.scaling.values)[None]
    scaled = oracle.scaling_adjust(sim, np.repeat(sc, P, 0), gidx, kind)
    mean = oracle.group_means(np.ascontiguousarray(scaled.T), np.zeros(P, np.int32), 1, 1).reshape(1, T).astype(np.float32)
    tnum = gr._time_index(cube(sim, dims)).numeric_ns
    trend = oracle.polydetrend_trend(mean, gidx, G, window, tnum, 1)
    np.testing.assert_allclose(out.trend.transpose(*dims).values, np.repeat(trend, P, 0), rtol=2e-6)
    tr = np.repeat(trend, P, 0).astype(np.float32)
    det = scaled - tr if kind == "+" else scaled * (np.float32(1) / tr)
    a = oracle.qm_adjust(det, np.repeat(af, P, 0), np.repeat(hq, P, 0), gidx, "nearest", "constant", kind)
    exp = a + tr if kind == "+" else a * tr
    np.testing.assert_allclose(out.scen.transpose(*dims).values, exp, rtol=2e-5, atol=2e-4)


def test_add_dims_error():
    """reference tests/test_adjustment.py:119-151"""
    rng = np.random.default_rng(1)
    x = cube(series(rng, (365,)), ("time",))
    with pytest.raises(ValueError, match=r"`add_dims` argument needs to be a dimension in one of the input datasets."):
        xs.EmpiricalQuantileMapping.train(x, x, group=xs.Grouper("time.month", add_dims=["dim_not_in_ref_or_hist"]))


def test_scaling_and_loci_pool_members():
    """reference tests/test_adjustment.py:196-203 (LOCI add_dims=["member"]) and :258-276 (Scaling add_dims=["lon"])"""
    rng = np.random.default_rng(2)
    P, T = 4, 3 * 365
    x = uniform(loc=0, scale=3).ppf(rng.random((P, T))).astype(np.float32)
    dims = ("member", "time")
    hist = cube(x, dims, "kg m-2 s-1")
    ref = cube(x * 2, dims, "kg m-2 s-1")
    gr = xs.Grouper("time.month", add_dims=["member"])
    sca = xs.Scaling.train(ref, hist, group=gr, kind="*")
    assert "member" not in sca.ds.af.dims
    np.testing.assert_allclose(sca.ds.af.values, 2, rtol=1e-5)
    np.testing.assert_allclose(sca.adjust(hist).transpose(*dims).values, x * 2, rtol=1e-5)
    loci = xs.LOCI.train(ref, hist, group=gr, thresh="2 kg m-2 s-1")
    assert "member" not in loci.ds.af.dims
    gidx, G = gr.group_index(hist)
    eaf, eth = oracle.loci_train(np.concatenate(list(x * 2))[None], np.concatenate(list(x))[None], np.tile(gidx, P), G, 1, 2.0)
    np.testing.assert_allclose(loci.ds.af.values.ravel(), eaf.ravel(), rtol=1e-6)
    np.testing.assert_allclose(loci.ds.hist_thresh.values.ravel(), eth.ravel(), rtol=1e-6)
    assert loci.adjust(hist, interp="nearest").values.shape == (P, T) or loci.adjust(hist, interp="nearest").values.shape == (T, P)


@pytest.mark.parametrize("group", ["time", "time.month"])
def test_adapt_freq_add_dims(group):
    """reference tests/test_adjustment.py:500-549: `add_dims` are NOT pooled for the frequency adaptation (_processing.py:85-86) —
    P0_ref / P0_hist / pth of the training keep the extra dimension, and dP0 of the training with add_dims=["point"] equals the
    bare adapt_freq run point by point; the quantiles are pooled as ever."""
    rng = np.random.default_rng(11)
    P, T = 3, 4 * 365
    ref = rng.gamma(2.0, 3.0, (P, T)).astype(np.float32).clip(2, None)
    hist = rng.gamma(2.0, 3.0, (P, T)).astype(np.float32).clip(2, None)
    it = np.arange(0, 30, 2)
    hist[1, it] = np.arange(len(it)) / len(it)  # below the threshold
    ref[:, :5] = 0
    ref[2, :10] = 0
    dims = ("point", "time")
    r, h = cube(ref, dims, "kg m-2 d-1"), cube(hist, dims, "kg m-2 d-1")
    adapted, _, dP0 = xs.processing.adapt_freq(r, h, thresh="1 kg m-2 d-1", group=group)   # bare: every point on its own
    dqm = xs.DetrendedQuantileMapping.train(r, h, kind="*", group=xs.Grouper(group, add_dims=["point"]), adapt_freq_thresh="1 kg m-2 d-1", nquantiles=10)
    assert "point" in dqm.ds.P0_ref.dims and "point" in dqm.ds.P0_hist.dims and "point" in dqm.ds.pth.dims
    assert "point" not in dqm.ds.af.dims and "point" not in dqm.ds.hist_q.dims
    P0_hist, P0_ref = dqm.ds.P0_hist.transpose("point", ...).values, dqm.ds.P0_ref.transpose("point", ...).values
    with np.errstate(divide="ignore", invalid="ignore"):
        dP0_tr = np.where(P0_hist == 0, np.nan, (P0_hist - P0_ref) / P0_hist)
    np.testing.assert_allclose(dP0_tr.reshape(P, -1), dP0.transpose("point", ...).values.reshape(P, -1), rtol=1e-6)
    # the per-point statistics themselves, from the oracle on each member alone
    gidx, G = xs.Grouper(group).group_index(r)
    for p_ in range(P):
        _, pth, _, p0r, p0h = oracle.adapt_freq(hist[p_][None], ref[p_][None], np.asarray(gidx, np.int32), G, 1, 1.0, seed=0)
        np.testing.assert_allclose(dqm.ds.P0_ref.transpose("point", ...).values[p_].ravel(), p0r.ravel(), rtol=1e-6)
        np.testing.assert_allclose(dqm.ds.P0_hist.transpose("point", ...).values[p_].ravel(), p0h.ravel(), rtol=1e-6)
        np.testing.assert_allclose(dqm.ds.pth.transpose("point", ...).values[p_].ravel(), pth.ravel(), rtol=1e-6, equal_nan=True)
    # no adjustment, just the frequency adaptation (the reference's check): what was to change changed, nothing else
    dqm.ds["af"] = dqm.ds["af"] * 0 + 1
    dqm.ds["scaling"] = dqm.ds["scaling"] * 0 + 1
    only = dqm.adjust(h).transpose(*dims).values
    ad = adapted.transpose(*dims).values
    np.testing.assert_allclose(only[0], hist[0], rtol=1e-5)            # point 0: nothing below the threshold in hist
    for p_, n_keep in ((1, 5), (2, 10)):
        c1 = np.abs(ad[p_, it] - hist[p_, it]) < 1e-6
        c2 = np.abs(only[p_, it] - hist[p_, it]) < 1e-6
        np.testing.assert_array_equal(c1, c2)
        np.testing.assert_allclose((ad[p_, it] - hist[p_, it])[:n_keep], 0)


@pytest.mark.parametrize("typ", ["ref", "hist"])
def test_add_dims_only_one_dataset(typ):
    """reference tests/test_adjustment.py:551-582: the extra dimension in only one of ref / hist — the other is broadcast along
    it for the frequency adaptation (_processing.py:104-106)"""
    rng = np.random.default_rng(12)
    T, P = 4 * 365, 3
    ref = rng.gamma(2.0, 3.0, T).astype(np.float32).clip(2, None)
    hist = rng.gamma(2.0, 3.0, T).astype(np.float32).clip(2, None)
    it = np.arange(0, 30, 2)
    hist[it] = np.arange(len(it)) / len(it)
    ref[:5] = 0
    r, h = cube(ref, ("time",), "kg m-2 d-1"), cube(hist, ("time",), "kg m-2 d-1")
    rp = cube(np.tile(ref, (P, 1)), ("point", "time"), "kg m-2 d-1")
    hp = cube(np.tile(hist, (P, 1)), ("point", "time"), "kg m-2 d-1")
    a, b = (rp, h) if typ == "ref" else (r, hp)
    dqm = xs.DetrendedQuantileMapping.train(a, b, kind="*", group=xs.Grouper("time", add_dims=["point"]), adapt_freq_thresh="1 kg m-2 d-1",
                                            jitter_under_thresh_value="0.01 kg m-2 d-1", nquantiles=10)
    assert "point" in dqm.ds.P0_ref.dims and dqm.ds.P0_ref.transpose("point", ...).values.shape[0] == P
    # every member saw the same data: the same statistics, those of the plain training
    plain = xs.DetrendedQuantileMapping.train(r, h, kind="*", group="time", adapt_freq_thresh="1 kg m-2 d-1", nquantiles=10)
    for k in ("P0_ref", "P0_hist"):
        v = dqm.ds[k].transpose("point", ...).values.reshape(P, -1)
        np.testing.assert_allclose(v, np.tile(plain.ds[k].values.reshape(1, -1), (P, 1)), rtol=1e-6)
    adj = dqm.adjust(b).values
    assert np.isfinite(adj).all()
