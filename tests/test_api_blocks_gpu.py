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
    # ... and the blocked result itself is held to the ORACLE on cells of the first and the last block (not only to the
    # one-piece device call); the grouped-linear case is checked against scipy in tests/test_interp_linear_gpu.py
    if "interp" not in adj_kw:
        import oracle
        from conftest import assert_close, assert_close_or_tie, assert_dqm_close_or_flip

        r, h, s_ = (a.reshape(-1, a.shape[-1]) for a in _data(kind, (7, 5, 3650)))
        sel = [0, 1, 17, 34]
        T = r.shape[1]
        doy = (np.arange(T) % 365).astype(np.int32)
        q = np.asarray(objb.ds["quantiles"].values, np.float32)
        got = scenb.transpose("lat", "lon", "time").values.reshape(-1, T)[sel]
        gaf = objb.ds["af"].transpose("lat", "lon", "dayofyear", "quantiles").values.reshape(35, 365, -1)[sel]
        if clsname == "DetrendedQuantileMapping":
            tnum = np.asarray(xs.date_range("2000-01-01", T, "noleap").numeric_ns, np.float64)
            oaf, ohq, osc = oracle.dqm_train(r[sel], h[sel], doy, 365, 31, q, kind)
            assert_close(gaf, oaf, rtol=3e-7, atol=3e-5, what="blocked dqm af vs oracle")
            oscen, otrend = oracle.dqm_adjust(s_[sel], oaf, ohq, osc, doy, 31, tnum, 1, "nearest", "constant", kind)
            assert_dqm_close_or_flip(got, oscen, s_[sel], osc, otrend, ohq, oaf, doy, kind, rtol=1e-5, what="blocked dqm scen vs oracle")
        else:
            oaf, ohq = oracle.eqm_train(r[sel], h[sel], doy, 365, 31, q, kind)
            np.testing.assert_array_equal(gaf, oaf)
            if clsname == "QuantileDeltaMapping":
                oscen, osq = oracle.qdm_adjust(s_[sel], oaf, q, doy, 31, False, "nearest", "constant", kind)
                assert_close_or_tie(got, oscen, s_[sel], q, oaf, doy, kind, rtol=1e-6, what="blocked qdm scen vs oracle", newx=osq, max_frac=5e-3)
            else:
                oscen = oracle.qm_adjust(s_[sel], oaf, ohq, doy, "nearest", "constant", kind)
                assert_close_or_tie(got, oscen, s_[sel], ohq, oaf, doy, kind, rtol=1e-6, what="blocked eqm scen vs oracle")
