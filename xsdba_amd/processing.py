"""Pre-processing functions of the quantile-mapping path (xsdba.processing): ``adapt_freq`` (:46-122),
``jitter`` / ``jitter_under_thresh`` / ``jitter_over_thresh`` (:124-257), each one call into the HIP library.

Random numbers: the reference draws from numpy's global generator.  Here every call takes ONE integer from that same
global generator (so ``np.random.seed(...)`` makes a script reproducible, as with the reference) and uses it as the
seed of the device's counter-based generator (include/xsdba_hip.h, f1)."""
from __future__ import annotations

import numpy as np

from . import hip
from .base import Grouper, parse_group
from .labeled import Coordinate, DataArray, as_dataarray
from .utils import cell_time_pair, cell_time_view

__all__ = ["adapt_freq", "jitter", "jitter_over_thresh", "jitter_under_thresh"]


def _draw_seed() -> int:
    """One seed for the device's counter-based generator, from numpy's global generator.  Inside a block worker of
    ``map_blocks`` (several devices) the blocks run in threads whose order is not fixed: there ONE base is drawn — by whichever
    block asks first — and a block's k-th seed is a function of (base, block number, k), so that ``np.random.seed(...)`` still
    fixes every draw."""
    from ._lib import _tls

    st = getattr(_tls, "seeds", None)
    if st is None:
        return int(np.random.randint(0, 2**31 - 1)) * 2654435761 + int(np.random.randint(0, 2**31 - 1))
    shared, block = st["shared"], st["block"]
    with shared["lock"]:
        if shared["base"] is None:
            shared["base"] = int(np.random.randint(0, 2**31 - 1)) * 2654435761 + int(np.random.randint(0, 2**31 - 1))
    k = st["k"]
    st["k"] = k + 1
    x = (shared["base"] ^ (block * 0x9E3779B97F4A7C15) ^ (k * 0xC2B2AE3D27D4EB4F)) & 0xFFFFFFFFFFFFFFFF
    x ^= x >> 33; x = (x * 0xFF51AFD7ED558CCD) & 0xFFFFFFFFFFFFFFFF; x ^= x >> 33   # (a bijective mix: distinct (block, k) -> distinct seeds)
    return int(x >> 1)


def quantity_to_float(q, like: DataArray | None = None, name: str = "threshold") -> float | None:
    """'1 mm d-1' -> the number in the units of ``like`` (units.convert_units_to :226-266; xsdba_amd/units.py); plain
    numbers pass through."""
    if q is None:
        return None
    if isinstance(q, (int, float, np.integer, np.floating)):
        return float(q)
    from .units import conversion, str2quantity

    val, units = str2quantity(q)
    have = str(like.attrs.get("units", "")).strip() if like is not None else ""
    if units and have:
        try:
            f, o = conversion(units, have)
        except ValueError as err:
            raise ValueError(f"{name} {q!r} cannot be expressed in the units of the data ({have!r}): {err}") from err
        val = val * f + o
    return val


def _series(da: DataArray, vals, cell_dims, cell_shape, dim, attrs=None):
    coords = {k: c for k, c in da.coords.items() if k in cell_dims or k == dim}
    out = DataArray(np.asarray(vals).reshape(cell_shape + (vals.shape[-1],)), cell_dims + (dim,), coords, dict(attrs or da.attrs), da.name)
    return out.transpose(*da.dims)


def _broadcast_along(a: DataArray, other: DataArray, dims):
    """`a` with the dimensions of `dims` it lacks taken from `other` (size, coordinate), in front — xr.broadcast for those"""
    add = [d for d in dims if d not in a.dims and d in other.dims]
    if not add:
        return a
    shape = tuple(other.shape[other.dims.index(d)] for d in add)
    v = np.broadcast_to(np.asarray(a.values), shape + tuple(a.shape))
    coords = dict(a.coords)
    coords.update({d: other.coords[d] for d in add if d in other.coords})
    return DataArray(np.ascontiguousarray(v), tuple(add) + tuple(a.dims), coords, dict(a.attrs), a.name)


def jitter(x, lower=None, upper=None, minimum=None, maximum=None) -> DataArray:
    """Replace values under a threshold and values above another by a uniform random noise (processing.py:178-257):
    values < lower -> U(minimum, lower) (minimum excluded, default 0); values >= upper -> U(upper, maximum)."""
    x = as_dataarray(x)
    lo, up = quantity_to_float(lower, x, "lower"), quantity_to_float(upper, x, "upper")
    mn, mx = quantity_to_float(minimum, x, "minimum"), quantity_to_float(maximum, x, "maximum")
    if up is not None and mx is None:
        raise ValueError("If 'upper' is given, so must 'maximum'.")
    dim = "time" if "time" in x.dims else x.dims[-1]
    x2, cell_dims, cell_shape = cell_time_view(x, dim)
    dt = np.result_type(x2.dtype, np.float32)
    out = hip.jitter(np.ascontiguousarray(x2, dt), lower=lo, minimum=mn, upper=up, maximum=mx, seed=_draw_seed())
    res = _series(x, out, cell_dims, cell_shape, dim)
    return res.astype(x.dtype) if res.dtype != x.dtype and np.issubdtype(x.dtype, np.floating) else res


def jitter_under_thresh(x, thresh) -> DataArray:
    """Replace values smaller than threshold by a uniform random noise (processing.py:124-149)."""
    return jitter(x, lower=thresh, upper=None, minimum=None, maximum=None)


def jitter_over_thresh(x, thresh, upper_bnd) -> DataArray:
    """Replace values greater than threshold by a uniform random noise (processing.py:152-175)."""
    return jitter(x, lower=None, upper=thresh, minimum=None, maximum=upper_bnd)


@parse_group
def adapt_freq(ref, sim, *, group: Grouper | str, thresh="0 mm d-1"):
    """Adapt frequency of values under thresh of `sim`, in order to match ref (processing.py:46-122, Themeßl et al. 2012).
    Returns (sim_ad, pth, dP0)."""
    ref, sim = as_dataarray(ref), as_dataarray(sim)
    th = quantity_to_float(thresh, ref, "thresh")
    dt = np.result_type(ref.dtype, sim.dtype, np.float32)
    gidx, G = group.group_index(sim)
    if group.add_dims:
        # _processing.py:85-86: "ADD_DIMS are not pooled with other dims for frequency adaptation" — the probabilities, pth and
        # the ranks are taken along time (and the window) for every member on its own; pth / dP0 keep the extra dimensions
        # (_processing.py:20-26, tests/test_processing.py:166-182: "lat" in pth.dims).  A dataset that lacks one of them is
        # broadcast along it (:104-106).  Members are therefore cells here.
        extra = [d for d in group.add_dims if d in sim.dims or d in ref.dims]
        if not extra:
            raise ValueError("`add_dims` argument needs to be a dimension in one of the input datasets.")
        sim, ref = _broadcast_along(sim, ref, extra), _broadcast_along(ref, sim, extra)
    s2, r2, cell_dims, cell_shape = cell_time_pair(sim.astype(dt) if sim.dtype != dt else sim,
                                                   ref.astype(dt) if ref.dtype != dt else ref, group.dim)
    sim_ad, pth, dP0, _, _ = hip.adapt_freq(s2, r2, gidx, G, group.window, th, seed=_draw_seed())
    attrs = dict(sim.attrs)
    attrs["references"] = ("Themeßl et al. (2012), Empirical-statistical downscaling and error correction of regional climate "
                           "models and its impact on the climate change signal, Climatic Change, DOI 10.1007/s10584-011-0224-4.")
    out = _series(sim, sim_ad, cell_dims, cell_shape, group.dim, attrs)
    prop = group.prop
    crd = group.get_coordinate(sim)
    coords = {k: c for k, c in sim.coords.items() if k in cell_dims}
    coords[prop] = Coordinate(prop, crd.values)
    units = sim.attrs.get("units", "")
    pth_da = DataArray(pth.reshape(cell_shape + (G,)), cell_dims + (prop,), coords,
                       {"long_name": "Smallest value of the timeseries not corrected by frequency adaptation.", "units": units}, "pth")
    dP0_da = DataArray(dP0.reshape(cell_shape + (G,)), cell_dims + (prop,), coords,
                       {"long_name": f"Proportion of values smaller than {th} in the timeseries corrected by frequency adaptation"}, "dP0")
    return out, pth_da, dP0_da
