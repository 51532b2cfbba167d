"""NetCDF round trip of datasets (reference tests/test_adjustment.py:443-452: ``ds.to_netcdf(file)`` ->
``open_dataset(file)`` -> ``from_dataset``).  The CPU part checks the file layer; the GPU part the trained object."""
from __future__ import annotations

import numpy as np
import pytest
from scipy.io import netcdf_file

import xsdba_amd as xs
from xsdba_amd.labeled import DataArray, Dataset


def test_dataset_roundtrip(tmp_path):
    rng = np.random.default_rng(0)
    t = xs.date_range("2000-01-01", 400, "noleap")
    ds = Dataset({
        "af": DataArray(rng.standard_normal((3, 12, 5)).astype(np.float32), ("lat", "month", "quantiles"),
                        {"lat": np.array([1.5, 2.5, 3.5]), "month": np.arange(1, 13), "quantiles": np.linspace(0.1, 0.9, 5).astype(np.float32)},
                        {"kind": "+", "long_name": "adjustment factors"}),
        "scen": DataArray(rng.standard_normal((3, 400)), ("lat", "time"), {"lat": np.array([1.5, 2.5, 3.5]), "time": t.values}, {"units": "K"}),
        "flag": DataArray(np.array([True, False, True]), ("lat",), {"lat": np.array([1.5, 2.5, 3.5])}),
        "season_tab": DataArray(np.arange(4.0), ("season",), {"season": np.array(["DJF", "MAM", "JJA", "SON"])}),
    }, attrs={"group": "time.month", "group_window": 1, "_xsdba_adjustment": '{"py/object": "x"}'})
    f = tmp_path / "ds.nc"
    ds.to_netcdf(f)
    back = xs.open_dataset(f)
    assert set(back.data_vars) == set(ds.data_vars)
    for name in ds.data_vars:
        a, b = ds[name], back[name]
        assert a.dims == b.dims and a.values.dtype == b.values.dtype, name
        np.testing.assert_array_equal(a.values, b.values)
        assert {k: v for k, v in a.attrs.items()} == {k: v for k, v in b.attrs.items()}
    assert back.attrs["group"] == "time.month" and back.attrs["group_window"] == 1
    assert back.attrs["_xsdba_adjustment"] == '{"py/object": "x"}'
    assert back["scen"].coords["time"].index.equals(t) and back["scen"].coords["time"].index.calendar == "noleap"
    assert list(back["season_tab"].coords["season"].values) == ["DJF", "MAM", "JJA", "SON"]
    np.testing.assert_array_equal(back["af"].coords["month"].values, np.arange(1, 13))
    # what xarray's scipy engine expects: CF time units + calendar, classic format
    with netcdf_file(str(f), "r", mmap=False) as nc:
        assert nc.variables["time"].units == b"days since 1970-01-01 00:00:00" and nc.variables["time"].calendar == b"noleap"
        assert nc.variables["af"].dimensions == ("lat", "month", "quantiles")


def test_standard_calendar_times(tmp_path):
    t = xs.date_range("1999-12-25", 70, "standard")
    da = DataArray(np.arange(70.0), ("time",), {"time": t.values}, {"units": "K"}, name="tas")
    f = tmp_path / "da.nc"
    da.to_netcdf(f)
    back = xs.open_dataset(f)["tas"]
    np.testing.assert_array_equal(np.asarray(back.coords["time"].values).astype("datetime64[D]"), np.asarray(t.values).astype("datetime64[D]"))


@pytest.mark.gpu
def test_trained_object_from_file(tmp_path):
    """reference tests/test_adjustment.py:443-452 (test_cannon_and_from_ds)"""
    rng = np.random.default_rng(1)
    t = xs.date_range("2000-01-01", 5 * 365, "noleap")
    mk = lambda a: DataArray(a.astype(np.float32), ("lat", "time"), {"lat": np.arange(2), "time": t.values}, {"units": "kg m-2 d-1"})  # noqa: E731
    ref, hist, sim = (mk(rng.gamma(2.0, s, (2, t.size))) for s in (3.0, 4.0, 5.0))
    dqm = xs.DetrendedQuantileMapping.train(ref, hist, kind="*", group=xs.Grouper("time.dayofyear", 31), nquantiles=10)
    p = dqm.adjust(sim)
    f = tmp_path / "dqm.nc"
    dqm.ds.to_netcdf(f)
    dqm2 = xs.DetrendedQuantileMapping.from_dataset(xs.open_dataset(f))
    for name in dqm.ds.data_vars:
        np.testing.assert_array_equal(dqm.ds[name].values, dqm2.ds[name].values)
    assert repr(dqm2) == repr(dqm)
    np.testing.assert_array_equal(dqm2.adjust(sim).values, p.values)


def test_cf_time_decoding_general_epochs(tmp_path):
    """xarray writes 'days since <first timestamp>' or hours: any '<unit> since <date>' epoch decodes, the time of day survives
    a round trip, and the julian calendar is refused by name"""
    from scipy.io import netcdf_file

    from xsdba_amd.io import open_dataset, to_netcdf

    def write(path, values, units, calendar):
        with netcdf_file(str(path), "w", version=2) as f:
            f.createDimension("time", len(values))
            t = f.createVariable("time", "d", ("time",))
            t[:] = values
            t.units = units
            t.calendar = calendar
            v = f.createVariable("x", "f", ("time",))
            v[:] = np.arange(len(values), dtype=np.float32)

    p = tmp_path / "hours.nc"
    write(p, [12.0, 36.0, 60.0], "hours since 2001-02-27 00:00:00", "noleap")  # daily data stamped at noon
    ds = open_dataset(p)
    idx = ds["x"].coords["time"].index
    assert list(idx.month) == [2, 2, 3] and list(idx.day) == [27, 28, 1]  # noleap: no Feb 29
    np.testing.assert_allclose(idx.frac, 0.5)
    to_netcdf(ds, tmp_path / "again.nc")
    idx2 = open_dataset(tmp_path / "again.nc")["x"].coords["time"].index
    assert idx2.equals(idx) and np.allclose(idx2.numeric_ns, idx.numeric_ns)

    p = tmp_path / "std.nc"
    write(p, [0.5, 1.5], "days since 1999-12-31", "proleptic_gregorian")
    t = open_dataset(p)["x"].coords["time"].values
    assert t[0] == np.datetime64("1999-12-31T12:00:00") and t[1] == np.datetime64("2000-01-01T12:00:00")

    p = tmp_path / "jul.nc"
    write(p, [0.0], "days since 2000-01-01", "julian")
    with pytest.raises(NotImplementedError, match="julian"):
        open_dataset(p)
