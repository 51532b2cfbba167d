"""The drop-in API (EmpiricalQuantileMapping / DetrendedQuantileMapping / QuantileDeltaMapping .train -> .adjust,
reference adjustment.py:227-316) with the data resident on the GPU: large host inputs are uploaded once, the trained
tables never leave the device between train and adjust, a result is downloaded only when its values are read — and all of
that is bit-identical to the plain path (one library call per stage with host pointers)."""
from __future__ import annotations

import numpy as np
import pytest

import xsdba_amd as xs
from xsdba_amd import hip
from xsdba_amd.labeled import DataArray

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    hip.default_context(0)


def _cube(values, dims, units):
    values = np.asarray(values)
    coords = {"time": xs.date_range("2000-01-01", values.shape[dims.index("time")], "noleap").values}
    for d, n in zip(dims, values.shape):
        if d != "time":
            coords[d] = np.arange(n)
    return DataArray(values, dims, coords, {"units": units})


def _data(kind, shape=(6, 8, 3650), seed=3):
    rng = np.random.default_rng(seed)
    T = shape[-1]
    doy = np.arange(T) % 365
    if kind == "+":
        base = 280 - 20 * np.cos(2 * np.pi * (doy + 1) / 365)
        ref = base + 2 * rng.random(shape)
        hist = base + 1 + 2.1 * rng.random(shape)
        sim = base + 2 + 2.2 * rng.random(shape) + 0.1 * np.arange(T) / 365
    else:
        ref, hist, sim = rng.gamma(4, 7.5, shape), rng.gamma(8.15, 3.68, shape), rng.gamma(16, 2.63, shape)
    out = [a.astype(np.float32) for a in (ref, hist, sim)]
    out[1][0, 0, rng.random(T) < 0.02] = np.nan
    return out


CASES = [
    ("EmpiricalQuantileMapping", dict(group=xs.Grouper("time.dayofyear", window=31), nquantiles=20), "+", {}),
    ("EmpiricalQuantileMapping", dict(group="time.month", nquantiles=15), "*", {}),
    ("DetrendedQuantileMapping", dict(group=xs.Grouper("time.dayofyear", window=31), nquantiles=20), "+", {}),
    ("DetrendedQuantileMapping", dict(group="time", nquantiles=20), "*", {"interp": "linear"}),
    ("QuantileDeltaMapping", dict(group=xs.Grouper("time.dayofyear", window=31), nquantiles=20), "*", {"rank_window": False}),
    ("QuantileDeltaMapping", dict(group="time.month", nquantiles=20), "*", {}),
]


@pytest.mark.parametrize("clsname,train_kw,kind,adj_kw", CASES)
def test_device_resident_equals_plain_path(monkeypatch, clsname, train_kw, kind, adj_kw):
    cls = getattr(xs, clsname)
    units = "K" if kind == "+" else "mm/d"
    ref, hist, sim = (_cube(a, ("lat", "lon", "time"), units) for a in _data(kind))
    import warnings

    def run():
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            obj = cls.train(ref, hist, kind=kind, **train_kw)
            return obj, obj.adjust(sim, **adj_kw)

    monkeypatch.setenv("XSDBA_AMD_DEVICE_MIN_MB", "1e12")  # plain path: nothing is moved beforehand
    obj_h, scen_h = run()
    assert obj_h.ds["af"].device_data is None and scen_h.device_data is None
    monkeypatch.setenv("XSDBA_AMD_DEVICE_MIN_MB", "0")     # device-resident path
    obj_d, scen_d = run()
    assert obj_d.ds["af"].device_data is not None, "trained tables must stay on the device"
    assert obj_d.ds["hist_q"].device_data is not None
    assert scen_d.device_data is not None and scen_d._host is None, "the result is downloaded only when read"
    for name in ("af", "hist_q") + (("scaling",) if "scaling" in obj_h.ds else ()):
        np.testing.assert_array_equal(obj_d.ds[name].values, obj_h.ds[name].values, err_msg=name)
    assert scen_d.dims == scen_h.dims and scen_d.attrs["units"] == scen_h.attrs["units"]
    np.testing.assert_array_equal(scen_d.values, scen_h.values)
    # trained on the device, adjusting a small host series: the tables are used where they are
    monkeypatch.setenv("XSDBA_AMD_DEVICE_MIN_MB", "1e12")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        scen_m = obj_d.adjust(sim, **adj_kw)
    np.testing.assert_array_equal(scen_m.values, scen_h.values)


def test_device_arrays_in_and_time_major():
    """DeviceArray-backed inputs are taken as they are, in both layouts the kernels read"""
    ctx = hip.default_context()
    ref, hist, sim = _data("+", shape=(5, 7, 1825), seed=9)
    group = xs.Grouper("time.dayofyear", window=31)
    mk = lambda a, dims: _cube(a, dims, "K")  # noqa: E731
    exp = xs.EmpiricalQuantileMapping.train(mk(ref, ("lat", "lon", "time")), mk(hist, ("lat", "lon", "time")), group=group,
                                            nquantiles=20).adjust(mk(sim, ("lat", "lon", "time")))
    for dims, tr in ((("lat", "lon", "time"), lambda a: a), (("time", "lat", "lon"), lambda a: np.ascontiguousarray(np.moveaxis(a, -1, 0)))):
        def dev(a):
            h = mk(tr(a), dims)
            return DataArray(ctx.to_device(h.values), h.dims, h.coords, h.attrs)

        eqm = xs.EmpiricalQuantileMapping.train(dev(ref), dev(hist), group=group, nquantiles=20)
        scen = eqm.adjust(dev(sim))
        assert scen.device_data is not None
        got = scen.transpose("lat", "lon", "time").values
        np.testing.assert_array_equal(got, exp.values)


@pytest.mark.parametrize("clsname,adj_kw", [("EmpiricalQuantileMapping", {}), ("DetrendedQuantileMapping", {}), ("QuantileDeltaMapping", {"rank_window": False})])
def test_time_major_inputs_keep_the_same_dims_on_both_paths(monkeypatch, clsname, adj_kw):
    """(time, lat, lon) inputs — what xarray cubes are: `dims`, the axis order of `.values` and the trained tables do not depend
    on whether the data were moved to the device first (the size threshold XSDBA_AMD_DEVICE_MIN_MB), and follow the reference's
    template order (base.py:696-700)"""
    import warnings

    cls = getattr(xs, clsname)
    kind = "*" if clsname == "QuantileDeltaMapping" else "+"
    arrs = [np.ascontiguousarray(np.moveaxis(a, -1, 0)) for a in _data(kind, shape=(4, 5, 1825), seed=5)]
    ref, hist, sim = (_cube(a, ("time", "lat", "lon"), "K" if kind == "+" else "mm/d") for a in arrs)

    def run():
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            obj = cls.train(ref, hist, kind=kind, group=xs.Grouper("time.dayofyear", window=31), nquantiles=20)
            return obj, obj.adjust(sim, **adj_kw)

    monkeypatch.setenv("XSDBA_AMD_DEVICE_MIN_MB", "1e12")
    obj_h, scen_h = run()
    monkeypatch.setenv("XSDBA_AMD_DEVICE_MIN_MB", "0")
    obj_d, scen_d = run()
    assert scen_d.device_data is not None and scen_h.device_data is None
    assert scen_d.dims == scen_h.dims and scen_d.shape == scen_h.shape
    np.testing.assert_array_equal(scen_d.values, scen_h.values)
    for name in obj_h.ds.data_vars:
        assert obj_d.ds[name].dims == obj_h.ds[name].dims, name
        np.testing.assert_array_equal(obj_d.ds[name].values, obj_h.ds[name].values, err_msg=name)


def test_results_are_released_with_their_arrays():
    """owning DeviceArrays give their buffer back when they are collected (no leak over a loop that drops its results)"""
    import gc

    ctx = hip.default_context()
    x = np.zeros((64, 4096), np.float32)
    before = None
    for i in range(6):
        d = ctx.to_device(x)
        ptr = d.ptr
        del d
        gc.collect()
        if i == 0:
            before = ptr
    # the allocator hands the same block out again once it has been returned
    assert ptr == before or i > 0


def test_device_resident_ref_and_hist_with_swapped_cell_order(monkeypatch):
    """ref stored (lat, lon, time), hist stored (lon, lat, time), both on the device: the cells are paired by label
    (`hist.transpose(*ref.dims)` is a lazy relabel of a device buffer — the training must not pair them by buffer position)."""
    import warnings

    ref_v, hist_v, sim_v = _data("+", shape=(4, 6, 3650))
    monkeypatch.setenv("XSDBA_AMD_DEVICE_MIN_MB", "0")
    from xsdba_amd._lib import default_context

    ctx = default_context()
    ref = _cube(ref_v, ("lat", "lon", "time"), "K")
    hist_h = _cube(np.ascontiguousarray(hist_v.transpose(1, 0, 2)), ("lon", "lat", "time"), "K")
    hist_d = DataArray(ctx.to_device(hist_h.values), hist_h.dims, hist_h.coords, hist_h.attrs)
    ref_d = DataArray(ctx.to_device(ref.values), ref.dims, ref.coords, ref.attrs)
    sim = _cube(sim_v, ("lat", "lon", "time"), "K")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        want = xs.EmpiricalQuantileMapping.train(ref, _cube(hist_v, ("lat", "lon", "time"), "K"),
                                                 group=xs.Grouper("time.dayofyear", window=31), nquantiles=20, kind="+")
        got = xs.EmpiricalQuantileMapping.train(ref_d, hist_d, group=xs.Grouper("time.dayofyear", window=31), nquantiles=20, kind="+")
        s_want, s_got = want.adjust(sim), got.adjust(sim)
    for name in ("af", "hist_q"):
        np.testing.assert_array_equal(got.ds[name].transpose(*want.ds[name].dims).values, want.ds[name].values, err_msg=name)
    np.testing.assert_array_equal(s_got.transpose(*s_want.dims).values, s_want.values)


@pytest.mark.parametrize("nq", [15, 19, 21])
def test_adjust_with_one_wave_blocks_and_unaligned_tables(monkeypatch, nq):
    """kernels.hpp tab_stage: tables that start off a 16-byte boundary (15, 19, 21 nodes) are staged in shifted 16-byte
    pieces; the elements behind the last whole piece must be copied whatever the block size (XSDBA_HIP_ADJ_THREADS=64)."""
    rng = np.random.default_rng(5)
    C, T = 7, 3650
    doy = (np.arange(T) % 365).astype(np.int32)
    ref = (280 - 20 * np.cos(2 * np.pi * (doy + 1) / 365) + 2 * rng.random((C, T))).astype(np.float32)
    hist = (ref + 1 + rng.random((C, T))).astype(np.float32)
    sim = (hist + 1).astype(np.float32)
    q = ((np.arange(nq) + 0.5) / nq).astype(np.float32)
    af, hq = hip.eqm_train(ref, hist, doy, 365, 31, q, "+")
    want = hip.qm_adjust(sim, af, hq, doy, "nearest", "constant", "+")
    monkeypatch.setenv("XSDBA_HIP_ADJ_THREADS", "64")
    got = hip.qm_adjust(sim, af, hq, doy, "nearest", "constant", "+")
    np.testing.assert_array_equal(got, want)
