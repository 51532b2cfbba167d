"""NetCDF persistence of trained datasets and results (reference: ``obj.ds.to_netcdf(file)`` ->
``xr.open_dataset(file)`` -> ``Cls.from_dataset(ds)``, tests/test_adjustment.py:443-452, base.py:80-100).

The image has no netCDF4 / h5netcdf / zarr; scipy ships a NetCDF-3 reader / writer (``scipy.io.netcdf_file``), the format
xarray's own ``engine="scipy"`` reads and writes, so the files written here open with ``xarray.open_dataset`` and files
written by xarray in NETCDF3 format open here.  CF conventions as xarray applies them: time written as ``days since
1970-01-01`` with a ``calendar`` attribute and read from any ``<days|hours|minutes|seconds> since <date>`` epoch (the time
of day is kept), strings as char arrays with a ``string<n>`` dimension, int64 stored as int32 (NetCDF-3 has no
64-bit integers), the ``_xsdba_adjustment`` JSON and the other attributes as text attributes.
"""
from __future__ import annotations

import numpy as np
from scipy.io import netcdf_file

from .calendar import TimeIndex, _DPM, _canon
from .labeled import Coordinate, DataArray, Dataset

_EPOCH = "days since 1970-01-01 00:00:00"


def _encode(values):
    """numpy array -> (array NetCDF-3 can hold, extra dims, extra attrs)"""
    a = np.asarray(values)
    if a.dtype.kind in "US":
        s = a.astype("S")
        n = max(s.dtype.itemsize, 1)
        return s.view("S1").reshape(a.shape + (n,)), (f"string{n}",), {}
    if a.dtype == np.bool_:
        return a.astype(np.int8), (), {"dtype": "bool"}
    if a.dtype.kind in "iu" and a.dtype.itemsize == 8:
        if a.size and (a.max() > np.iinfo(np.int32).max or a.min() < np.iinfo(np.int32).min):
            raise ValueError("NetCDF-3 cannot hold 64-bit integers beyond the int32 range")
        return a.astype(np.int32), (), {}
    if a.dtype.kind in "iuf" and a.dtype.itemsize in (1, 2, 4, 8):
        return a, (), {}
    raise TypeError(f"cannot store dtype {a.dtype} in NetCDF-3")


def _attr(v):
    if isinstance(v, (str, bytes)):
        return v
    if isinstance(v, (bool, np.bool_)):
        return np.int8(v)
    if isinstance(v, (int, np.integer)):
        return np.int32(v)
    if isinstance(v, (float, np.floating)):
        return np.float64(v)
    if isinstance(v, (list, tuple, np.ndarray)) and len(v) and not isinstance(v[0], str):
        return np.asarray(v)
    return str(v)


def to_netcdf(ds: Dataset | DataArray, path) -> None:
    """Write `ds` (a Dataset, or a named DataArray) as a NetCDF-3 (64-bit offset) file."""
    if isinstance(ds, DataArray):
        ds = ds.to_dataset(name=ds.name or "__xarray_dataarray_variable__")
    with netcdf_file(str(path), "w", version=2) as f:
        for k, v in ds.attrs.items():
            setattr(f, k, _attr(v))
        sizes = ds.sizes
        for d, n in sizes.items():
            f.createDimension(d, int(n))
        made = set(sizes)

        def put(name, dims, values, attrs):
            arr, xdims, xattrs = _encode(values)
            for xd in xdims:
                if xd not in made:
                    f.createDimension(xd, int(xd[len("string"):]))
                    made.add(xd)
            var = f.createVariable(name, arr.dtype.newbyteorder("=") if arr.dtype.kind != "S" else "c", tuple(dims) + xdims)
            var[...] = arr
            for k, v in {**attrs, **xattrs}.items():
                setattr(var, k, _attr(v))

        for name, c in ds.coords.items():
            if name not in sizes:
                continue
            if c.index is not None:  # time axis -> CF days since the epoch in its calendar
                idx: TimeIndex = c.index
                put(name, (name,), idx.numeric_ns / 86400e9, {**c.attrs, "units": _EPOCH, "calendar": _cf_calendar(idx.calendar)})
            else:
                put(name, (name,), c.values, c.attrs)
        for name, da in ds.data_vars.items():
            put(name, da.dims, da.values, da.attrs)


def _cf_calendar(cal: str) -> str:
    c = _canon(cal)
    return {"standard": "proleptic_gregorian"}.get(c, c)


_UNIT_DAYS = {"days": 1.0, "day": 1.0, "d": 1.0, "hours": 1 / 24, "hour": 1 / 24, "hrs": 1 / 24, "hr": 1 / 24, "h": 1 / 24,
              "minutes": 1 / 1440, "minute": 1 / 1440, "mins": 1 / 1440, "min": 1 / 1440,
              "seconds": 1 / 86400, "second": 1 / 86400, "secs": 1 / 86400, "sec": 1 / 86400, "s": 1 / 86400}


def _parse_cf_time_units(units: str):
    """'<unit> since <YYYY-MM-DD[ HH:MM[:SS[.f]]]>' -> (length of the unit in days, (year, month, day), fraction of the
    epoch's day)"""
    unit, _, ref = units.partition(" since ")
    unit = unit.strip().lower()
    if unit not in _UNIT_DAYS:
        raise NotImplementedError(f"time units {units!r}: the unit must be one of days / hours / minutes / seconds")
    ref = ref.strip().replace("T", " ")
    date, _, clock = ref.partition(" ")
    try:
        y, m, d = (int(p) for p in date.split("-"))
        hh = mm = 0
        ss = 0.0
        if clock:
            clock = clock.split()[0].rstrip("Z")  # (a trailing time zone field other than UTC is not honoured)
            parts = clock.split(":")
            hh, mm, ss = int(parts[0]), (int(parts[1]) if len(parts) > 1 else 0), (float(parts[2]) if len(parts) > 2 else 0.0)
    except ValueError as err:
        raise NotImplementedError(f"cannot parse the reference date of time units {units!r}") from err
    return _UNIT_DAYS[unit], (y, m, d), (hh * 3600 + mm * 60 + ss) / 86400.0


def _decode_time(values: np.ndarray, units: str, calendar: str):
    """CF time decoding as xarray does it: any '<unit> since <date>' epoch, the time of day kept (datetime64[ns] for the
    standard calendar, year / month / day fields + a day fraction for the others)."""
    cal = _canon(calendar)
    if cal == "julian":
        raise NotImplementedError("the 'julian' calendar is not supported")
    scale, (y0, m0, d0), f0 = _parse_cf_time_units(units)
    v = np.asarray(values, np.float64)
    if cal == "standard":
        epoch = np.datetime64(f"{y0:04d}-{m0:02d}-{d0:02d}", "ns") + np.timedelta64(int(round(f0 * 86400e9)), "ns")
        return epoch + np.round(v * scale * 86400e9).astype("timedelta64[ns]")
    dpm = _DPM[cal]
    ylen = int(dpm.sum())
    days = v * scale + f0 + ((y0 - 1970) * ylen + int(dpm[: m0 - 1].sum()) + d0 - 1)
    d = np.floor(days + 1e-9).astype(np.int64)
    frac = np.clip(days - d, 0.0, None)
    frac[frac < 1e-9] = 0.0
    year = 1970 + d // ylen
    doy = d % ylen
    cum = np.concatenate([[0], np.cumsum(dpm)])
    month = np.searchsorted(cum, doy, side="right")
    day = doy - cum[month - 1] + 1
    return TimeIndex(year, month, day, calendar, frac=frac)


def open_dataset(path) -> Dataset:
    """Read a NetCDF-3 file written by :func:`to_netcdf` (or by xarray in NETCDF3 format) into a Dataset."""
    with netcdf_file(str(path), "r", mmap=False) as f:
        attrs = {k: (v.decode() if isinstance(v, bytes) else v) for k, v in f._attributes.items()}
        raw = {}
        for name, var in f.variables.items():
            a = np.array(var[...] if var.shape != () else var.getValue())
            dims = tuple(var.dimensions)
            vattrs = {k: (v.decode() if isinstance(v, bytes) else v) for k, v in var._attributes.items()}
            if a.dtype.kind == "S" and dims and dims[-1].startswith("string"):
                a = np.array([b"".join(row).decode() for row in a.reshape(-1, a.shape[-1])]).reshape(a.shape[:-1])
                dims = dims[:-1]
            if vattrs.pop("dtype", None) == "bool":
                a = a.astype(bool)
            if a.dtype.byteorder == ">":
                a = a.astype(a.dtype.newbyteorder("="))
            raw[name] = (dims, a, vattrs)
    coords = {}
    for name, (dims, a, vattrs) in raw.items():
        if dims == (name,):
            units = str(vattrs.get("units", ""))
            if " since " in units:
                cal = vattrs.pop("calendar", "standard")
                vattrs.pop("units")
                coords[name] = Coordinate(name, _decode_time(a, units, {"proleptic_gregorian": "standard", "gregorian": "standard"}.get(cal, cal)), vattrs)
            else:
                coords[name] = Coordinate(name, a, vattrs)
    out = {}
    for name, (dims, a, vattrs) in raw.items():
        if name in coords:
            continue
        out[name] = DataArray(a, dims, {d: coords[d] for d in dims if d in coords}, vattrs, name)
    ds = Dataset(out, coords={k: c for k, c in coords.items()}, attrs=attrs)
    return ds
