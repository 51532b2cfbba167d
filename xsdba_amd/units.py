"""Unit handling of the path's inputs (xsdba.units: ``convert_units_to`` :226-266, ``harmonize_units`` :269-321,
``_harmonize_units`` of the adjustment classes, adjustment.py:182-185).

The reference leans on pint + cf-xarray's "hydro" context; neither is available here, and the path only ever needs to
bring ref / hist / sim and the thresholds given as strings ("1 mm d-1") to common units.  This module parses the CF-style
unit strings climate data carry — products of powers of SI units with prefixes, ``/``, ``^`` or trailing exponents
(``kg m-2 s-1``, ``mm/d``, ``m s^-1``, ``W m-2``, ``degC``) — into (factor, offset, dimensions) and converts between units
of the same dimensions, with the hydro equivalence of a water mass per area and a water depth (1 kg m-2 = 1 mm).  Arrays
are converted on the device (one elementwise kernel); anything it cannot parse or convert raises, it never guesses.
"""
from __future__ import annotations

import re

import numpy as np

# name -> (factor to SI, {base dimension: power})
_BASE = {
    "m": (1.0, {"L": 1}), "g": (1e-3, {"M": 1}), "s": (1.0, {"T": 1}), "K": (1.0, {"K": 1}),
    "min": (60.0, {"T": 1}), "h": (3600.0, {"T": 1}), "hr": (3600.0, {"T": 1}), "hour": (3600.0, {"T": 1}),
    "d": (86400.0, {"T": 1}), "day": (86400.0, {"T": 1}), "days": (86400.0, {"T": 1}), "sec": (1.0, {"T": 1}), "second": (1.0, {"T": 1}),
    "yr": (365.25 * 86400.0, {"T": 1}), "year": (365.25 * 86400.0, {"T": 1}),
    "N": (1.0, {"M": 1, "L": 1, "T": -2}), "Pa": (1.0, {"M": 1, "L": -1, "T": -2}), "J": (1.0, {"M": 1, "L": 2, "T": -2}),
    "W": (1.0, {"M": 1, "L": 2, "T": -3}), "bar": (1e5, {"M": 1, "L": -1, "T": -2}),
    "1": (1.0, {}), "%": (0.01, {}), "percent": (0.01, {}),
    "in": (0.0254, {"L": 1}), "inch": (0.0254, {"L": 1}), "ft": (0.3048, {"L": 1}),
}
_PREFIX = {"k": 1e3, "h": 1e2, "da": 1e1, "d": 1e-1, "c": 1e-2, "m": 1e-3, "u": 1e-6, "µ": 1e-6, "M": 1e6, "G": 1e9}
_PREFIXABLE = {"m", "g", "s", "Pa", "J", "W", "N", "bar", "K"}
# temperature scales with an offset: name -> (factor, offset) to kelvin
_TEMP = {"degC": (1.0, 273.15), "°C": (1.0, 273.15), "celsius": (1.0, 273.15), "degree_Celsius": (1.0, 273.15), "degrees_Celsius": (1.0, 273.15),
         "deg_C": (1.0, 273.15), "C": (1.0, 273.15), "degF": (5.0 / 9.0, 459.67 * 5.0 / 9.0), "°F": (5.0 / 9.0, 459.67 * 5.0 / 9.0),
         "degK": (1.0, 0.0), "kelvin": (1.0, 0.0)}
_WATER_DENSITY = 1000.0  # kg m-3: the "hydro" context (units.py:50-84)


def _atom(tok: str):
    if tok in _BASE:
        return _BASE[tok]
    for p, f in _PREFIX.items():
        if tok.startswith(p) and tok[len(p):] in _PREFIXABLE:
            bf, dims = _BASE[tok[len(p):]]
            return bf * f, dims
    raise ValueError(f"unknown unit {tok!r}")


def parse(units: str):
    """units string -> (factor, offset, dims): value_SI = value * factor + offset"""
    u = " ".join(str(units).replace("**", "^").replace("·", " ").replace("*", " ").split())
    if u in _TEMP:
        f, o = _TEMP[u]
        return f, o, {"K": 1}
    factor, dims = 1.0, {}
    sign = 1
    for part in re.split(r"(/)", u):
        if part == "/":
            sign = -1
            continue
        for tok in part.split():
            m = re.fullmatch(r"([A-Za-z%°µ_1]+?)\^?(-?\d+)?", tok)
            if not m:
                raise ValueError(f"cannot parse the unit {tok!r} in {units!r}")
            name, power = m.group(1), int(m.group(2) or 1) * sign
            if name in _TEMP:  # a temperature difference inside a compound unit: the scale factor only
                f, d = _TEMP[name][0], {"K": 1}
            else:
                f, d = _atom(name)
            factor *= f ** power
            for k, v in d.items():
                dims[k] = dims.get(k, 0) + v * power
        if part != "/" and sign == -1:
            sign = -1  # everything after the first "/" stays in the denominator (CF: "mm/d", "W/m2")
    return factor, 0.0, {k: v for k, v in dims.items() if v}


def conversion(src: str, dst: str):
    """(factor, offset) with value_dst = value_src * factor + offset; raises ValueError when the dimensions differ
    (after the hydro equivalence water mass per area <-> water depth)"""
    if " ".join(str(src).split()) == " ".join(str(dst).split()):
        return 1.0, 0.0
    fs, os_, ds = parse(src)
    fd, od, dd = parse(dst)

    def hydro(f, d):  # bring M L-2 (T^k) to L (T^k): divide by the density of water
        if d.get("M", 0) == 1 and d.get("L", 0) == -2:
            d = dict(d)
            d.pop("M")
            d["L"] = 1
            return f / _WATER_DENSITY, {k: v for k, v in d.items() if v}
        return f, d

    if ds != dd:
        fs, ds = hydro(fs, ds)
        fd, dd = hydro(fd, dd)
    if ds != dd:
        raise ValueError(f"cannot convert {src!r} to {dst!r}: different dimensions")
    return fs / fd, (os_ - od) / fd


def str2quantity(q):
    """'1 mm d-1' -> (1.0, 'mm d-1'); numbers -> (value, '')"""
    if isinstance(q, (int, float, np.integer, np.floating)):
        return float(q), ""
    parts = str(q).strip().split(None, 1)
    try:
        return float(parts[0]), (parts[1].strip() if len(parts) > 1 else "")
    except ValueError as err:
        raise ValueError(f"cannot parse the quantity {q!r} (expected '<number> <units>')") from err


def convert_units_to(source, target):
    """units.convert_units_to (:226-266): `source` a quantity string / number (-> float in the target units) or a DataArray
    (-> DataArray in the target units, converted on the device); `target` a units string or a DataArray with ``units``."""
    from .labeled import DataArray, as_dataarray

    tu = target if isinstance(target, str) else str(as_dataarray(target).attrs.get("units", ""))
    if isinstance(source, (str, int, float, np.integer, np.floating)):
        val, su = str2quantity(source)
        if not su or not tu:
            return val
        f, o = conversion(su, tu)
        return val * f + o
    da = as_dataarray(source)
    su = str(da.attrs.get("units", ""))
    f, o = conversion(su, tu)
    if f == 1.0 and o == 0.0:
        out = da.copy()
    else:
        from . import hip
        from .utils import cell_time_view

        dim = "time" if "time" in da.dims else da.dims[-1]
        x2, cell_dims, cell_shape = cell_time_view(da, dim)
        dt = np.result_type(x2.dtype, np.float32)
        y = np.ascontiguousarray(x2, dt)
        ones = np.ones((y.shape[0], 1), dt)
        if f != 1.0:
            y = hip.scaling_adjust(y, ones * dt.type(f), None, "*")
        if o != 0.0:
            y = hip.scaling_adjust(y, ones * dt.type(o), None, "+")
        coords = {k: c for k, c in da.coords.items() if k in cell_dims or k == dim}
        out = DataArray(y.reshape(cell_shape + (y.shape[-1],)), cell_dims + (dim,), coords, dict(da.attrs), da.name).transpose(*da.dims)
    out.attrs["units"] = tu
    return out
