"""map_blocks cuts inputs larger than the block budget into blocks of cells (the role dask chunks play for the reference,
base.py:640-702): train and adjust through the drop-in API on a cube several times the (artificially small) budget give
exactly what the one-piece call gives — for both cube layouts, with blocks uploaded through the device-resident path or the
plain one."""
from __future__ import annotations

import warnings

import numpy as np
import pytest

import xsdba_amd as xs
from xsdba_amd import base, hip
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
            coords[d] = np.arange(n) * 2.5
    return DataArray(values, dims, coords, {"units": units})


def _data(kind, shape, seed=11):
    rng = np.random.default_rng(seed)
    T = shape[-1]
    doy = np.arange(T) % 365
    if kind == "+":
        b = 280 - 20 * np.cos(2 * np.pi * (doy + 1) / 365)
        arrs = [b + 2 * rng.random(shape), b + 1 + 2.1 * rng.random(shape), b + 2 + 2.2 * rng.random(shape) + 0.1 * np.arange(T) / 365]
    else:
        arrs = [rng.gamma(4, 7.5, shape), rng.gamma(8.15, 3.68, shape), rng.gamma(16, 2.63, shape)]
    return [a.astype(np.float32) for a in arrs]


CASES = [
    ("EmpiricalQuantileMapping", dict(group=xs.Grouper("time.dayofyear", window=31), nquantiles=20), "+", {}),
    ("DetrendedQuantileMapping", dict(group=xs.Grouper("time.dayofyear", window=31), nquantiles=20), "+", {}),
    ("QuantileDeltaMapping", dict(group=xs.Grouper("time.dayofyear", window=31), nquantiles=20), "*", {}),
    ("EmpiricalQuantileMapping", dict(group="time.month", nquantiles=15), "*", {"interp": "linear"}),
]


@pytest.mark.parametrize("dims", [("lat", "lon", "time"), ("time", "lat", "lon")])
@pytest.mark.parametrize("min_mb", ["0", "1e12"])
@pytest.mark.parametrize("clsname,train_kw,kind,adj_kw", CASES)
def test_blocks_equal_one_piece(monkeypatch, clsname, train_kw, kind, adj_kw, dims, min_mb):
    cls = getattr(xs, clsname)
    units = "K" if kind == "+" else "mm/d"
    arrs = _data(kind, (7, 5, 3650))
    if dims[0] == "time":
        arrs = [np.ascontiguousarray(np.moveaxis(a, -1, 0)) for a in arrs]
    ref, hist, sim = (_cube(a, dims, units) for a in arrs)

    def run():
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            obj = cls.train(ref, hist, kind=kind, **train_kw)
            return obj, obj.adjust(sim, **adj_kw)

    monkeypatch.setenv("XSDBA_AMD_DEVICE_MIN_MB", min_mb)
    monkeypatch.setenv("XSDBA_AMD_BLOCK_MAX_MB", "1e9")
    obj1, scen1 = run()
    calls = []
    real = base._isel_block
    monkeypatch.setattr(base, "_isel_block", lambda ds, dim, sl: (calls.append((dim, sl)), real(ds, dim, sl))[1])
    monkeypatch.setenv("XSDBA_AMD_BLOCK_MAX_MB", "0.3")  # two series of 0.5 MB per train call: four blocks of lat; more for adjust
    objb, scenb = run()
    assert len(calls) >= 5 and {d for d, _ in calls} == {"lat"}, calls
    for name in obj1.ds.data_vars:
        a, b = obj1.ds[name], objb.ds[name]
        np.testing.assert_array_equal(b.transpose(*a.dims).values, a.values, err_msg=name)
        assert objb.ds[name].coords["lat"].equals(obj1.ds[name].coords["lat"]) if "lat" in a.dims else True
    np.testing.assert_array_equal(scenb.transpose(*scen1.dims).values, scen1.values)
    assert scenb.attrs["units"] == scen1.attrs["units"]
