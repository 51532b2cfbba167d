"""Array helpers of the hot path with the reference's names (xsdba.utils, utils.py:31-32, 130-177,
251-281, 284-314, 408-513, 573-638).  Numerics run in the HIP library; these wrappers only shape the
labelled inputs/outputs."""
from __future__ import annotations

from warnings import warn

import numpy as np

from . import hip
from .base import Grouper, parse_group
from .labeled import Coordinate, DataArray, as_dataarray

MULTIPLICATIVE = "*"
ADDITIVE = "+"


def equally_spaced_nodes(n: int, eps: float | None = None) -> np.ndarray:
    """Nodes with `n` equally spaced points within [0, 1], optionally with two end-points (utils.py:251-281)."""
    dq = 1 / n / 2
    q = np.linspace(dq, 1 - dq, n)
    if eps is None:
        return q
    return np.insert(np.append(q, 1 - eps), 0, eps)


def get_correction(x, y, kind: str):
    """Additive or multiplicative correction factors (utils.py:130-143)."""
    if kind == ADDITIVE:
        out = y - x
    elif kind == MULTIPLICATIVE:
        out = y / x
    else:
        raise ValueError("kind must be + or *.")
    if isinstance(out, DataArray):
        out.attrs["kind"] = kind
    return out


def apply_correction(x, factor, kind: str | None = None):
    """Apply correction factors (utils.py:146-162)."""
    kind = kind or getattr(factor, "attrs", {}).get("kind", None)
    if kind == ADDITIVE:
        return x + factor
    if kind == MULTIPLICATIVE:
        return x * factor
    raise ValueError("kind must be `+` or `*`.")


def invert(x, kind: str | None = None):
    """-x or 1/x (utils.py:165-177)."""
    kind = kind or getattr(x, "attrs", {}).get("kind", None)
    if kind == ADDITIVE:
        return -x
    if kind == MULTIPLICATIVE:
        return 1 / x
    raise ValueError


def add_cyclic_bounds(da, att: str, cyclic_coords: bool = True):
    """Prepend the last slice and append the first along `att` (utils.py:284-314)."""
    da = as_dataarray(da)
    ax = da.dims.index(att)
    v = da.values
    data = np.concatenate([np.take(v, [-1], axis=ax), v, np.take(v, [0], axis=ax)], axis=ax)
    crd = np.asarray(da.coords[att].values)
    if cyclic_coords:
        vals = np.concatenate([crd[-1:], crd, crd[:1]])
    else:
        vals = np.concatenate([[crd[0] - (crd[1] - crd[0])], crd, [crd[-1] + (crd[-1] - crd[-2])]])
    coords = dict(da.coords)
    coords[att] = Coordinate(att, vals, da.coords[att].attrs)
    return DataArray(data, da.dims, coords, da.attrs, da.name)


# ---- [cell, time] views of labelled arrays ------------------------------------------------------------
def cell_time_view(da: DataArray, dim: str = "time"):
    """Logical ``[cell, time]`` 2-D view of `da` (no copy for C-ordered (time, cells...) or (cells..., time)
    arrays: the library reads both layouts), with the cell dims and their shape to fold results back."""
    ax = da.dims.index(dim)
    cell_dims = tuple(d for d in da.dims if d != dim)
    dev = getattr(da, "device_data", None)
    if dev is not None and getattr(da, "_host", None) is None:
        # device-resident data: (cells..., time) is a reshape of the buffer, (time, cells...) the time-major view of it.
        # The buffer's own dimension order counts (`da` may be a relabelled — lazily transposed — view of it): the views are
        # only handed out when the buffer orders the cells the way the labels do, because callers pair the cells of several
        # arrays by position after `x.transpose(*other.dims)`; anything else takes the labelled-order host path below.
        pdims = tuple(da.device_dims)
        pax = pdims.index(dim)
        if tuple(d for d in pdims if d != dim) == cell_dims:
            shape = tuple(dev.shape)
            cell_shape = shape[:pax] + shape[pax + 1:]
            n_cell, n_time = int(np.prod(cell_shape, dtype=np.int64)), shape[pax]
            if pax == len(shape) - 1:
                return dev.reshape(n_cell, n_time), cell_dims, cell_shape
            if pax == 0 and getattr(dev, "cell_time_strides", None) is None:
                from ._lib import DeviceArray

                v = DeviceArray.from_ptr(dev.ctx, dev.ptr, (n_time, n_cell), dev.dtype)
                v._base = dev
                v.logical_shape, v.cell_time_strides = (n_cell, n_time), (1, n_cell)
                return v, cell_dims, cell_shape
    v = np.moveaxis(da.values, ax, -1)
    cell_shape = v.shape[:-1]
    return v.reshape(-1, v.shape[-1]), cell_dims, cell_shape


def _is_dev(v) -> bool:
    return type(v).__name__ == "DeviceArray"


def _upload_like(host2d, like):
    """the host ``[cell, time]`` block on the device, laid out like the device view `like` (cell-major, or the time-major view
    `cell_time_view` hands out): the library takes the series of one call from ONE kind of memory in ONE layout"""
    from ._lib import DeviceArray

    strides = getattr(like, "cell_time_strides", None)
    if strides is None:
        return like.ctx.to_device(np.ascontiguousarray(host2d, like.dtype))
    n_cell, n_time = host2d.shape
    buf = like.ctx.to_device(np.ascontiguousarray(np.asarray(host2d, like.dtype).T))
    v = DeviceArray.from_ptr(buf.ctx, buf.ptr, (n_time, n_cell), buf.dtype)
    v._base = buf
    v.logical_shape, v.cell_time_strides = (n_cell, n_time), (1, n_cell)
    return v


def cell_time_pair(a: DataArray, b: DataArray, dim: str = "time"):
    """``[cell, time]`` views of `a` and of `b` with b's cells in a's (labelled) order, both in the same memory and layout:
    (a2, b2, cell_dims, cell_shape).  The arrays whose cells are paired by position (ref / hist at training, sim / ref in
    adapt_freq) come through here: a device-resident array stored in another cell order, or one array on the device and one
    on the host, is brought to the layout of the other instead of being paired by buffer position."""
    a2, cell_dims, cell_shape = cell_time_view(a, dim)
    b2, _, _ = cell_time_view(b.transpose(*a.dims), dim)
    da, db = _is_dev(a2), _is_dev(b2)
    if da and db and getattr(a2, "cell_time_strides", None) != getattr(b2, "cell_time_strides", None):
        h = b2.download().reshape(b2.shape)
        b2, db = (h.T if getattr(b2, "cell_time_strides", None) is not None else h), False
    if da and not db:
        b2 = _upload_like(b2, a2)
    elif db and not da:
        a2 = _upload_like(a2, b2)
    return a2, b2, cell_dims, cell_shape


def table_view(tab: DataArray, cell_dims, prop: str, last: str | None = "quantiles", cell_shape=None):
    """[cell, group(, quantile)] contiguous view of a trained table with the cells ordered like `cell_dims`.
    Cell dimensions the table does not have (the members pooled by ``Grouper(add_dims=...)`` at training) are broadcast,
    their sizes taken from `cell_shape`."""
    tail = [prop] + ([last] if last else [])
    dev = getattr(tab, "device_data", None)
    if dev is not None and getattr(tab, "_host", None) is None and tuple(tab.device_dims) == tuple(cell_dims) + tuple(tail):
        # a device-resident table already laid out (cells..., group(, quantile)): a reshape of the buffer, nothing moves
        n_cell = int(np.prod(tab.shape[:len(cell_dims)], dtype=np.int64))
        return dev.reshape((n_cell,) + tuple(tab.shape[len(cell_dims):]))
    missing = [d for d in cell_dims if d not in tab.dims]
    if missing and cell_shape is None:
        raise ValueError(f"trained table lacks dimension(s) {missing} present on the data")
    have = [d for d in cell_dims if d in tab.dims]
    t = tab.transpose(*have, *tail)
    v = np.asarray(t.values)
    if missing:
        shape = [n if d in have else 1 for d, n in zip(cell_dims, cell_shape)] + list(v.shape[len(have):])
        full = [n for n in cell_shape] + list(v.shape[len(have):])
        v = np.broadcast_to(v.reshape(shape), full)
    return np.ascontiguousarray(v).reshape((-1,) + tuple(v.shape[len(cell_dims):]))


@parse_group
def interp_on_quantiles(newx, xq, yq, *, group: str | Grouper = "time", method: str = "linear",
                        extrapolation: str = "constant"):
    """Interpolate values of yq on new values of x (utils.py:408-513).

    Ungrouped: 1-D scipy-interp1d semantics ('nearest' | 'linear').  Grouped: the reference's 2-D griddata path:
    ``method="nearest"`` (true 2-D nearest neighbour) and ``"linear"`` (Delaunay triangulation of every cell's nodes +
    piecewise-linear interpolant, csrc/lin2d.hpp) or ``"cubic"`` (the Clough-Tocher interpolant on the same triangulation).
    Ungrouped 'cubic' is interp1d's not-a-knot spline.
    """
    newx, xq, yq = as_dataarray(newx), as_dataarray(xq), as_dataarray(yq)
    dim, prop = group.dim, group.prop
    if method not in ("nearest", "linear", "cubic"):
        raise NotImplementedError(f"interp_on_quantiles(method={method!r}) with group {group.name!r} is not implemented on the HIP path")
    x2, cell_dims, cell_shape = cell_time_view(newx, dim)
    if prop == "group":
        if "group" in xq.dims:
            xq = xq.squeeze("group")
        if "group" in yq.dims:
            yq = yq.squeeze("group")
        xt = table_view(xq.expand_dims(group=[1]), cell_dims, "group")
        yt = table_view(yq.expand_dims(group=[1]), cell_dims, "group")
        gidx = None
    else:
        crd = group.get_coordinate(newx)
        if prop not in xq.dims:
            xq = xq.expand_dims({prop: crd.values})
        if prop not in yq.dims:
            yq = yq.expand_dims({prop: crd.values})
        for d in cell_dims:
            if d not in xq.dims:
                xq = xq.expand_dims({d: newx.sizes[d]})
            if d not in yq.dims:
                yq = yq.expand_dims({d: newx.sizes[d]})
        xt, yt = table_view(xq, cell_dims, prop), table_view(yq, cell_dims, prop)
        gidx, _ = group.group_index(newx)
    dt = np.result_type(yt.dtype, np.float32)
    if gidx is not None and method in ("linear", "cubic"):
        # the reference's griddata path: Delaunay triangulation of the (xq, group coordinate) nodes per cell (csrc/lin2d.hpp)
        c = (np.arange(4) if prop == "season" else np.asarray(crd.values)).astype(np.float64)
        gcoord = np.concatenate([[c[0] - (c[1] - c[0])], c, [c[-1] + (c[-1] - c[-2])]])
        newg = np.asarray(group.get_index(newx, interp=True).values, np.float64)
        out = hip.qm_adjust_linear2d(x2.astype(dt, copy=False), yt.astype(dt, copy=False), xt.astype(dt, copy=False), gcoord, newg,
                                     extrapolation, kind=None, cubic=method == "cubic")
        coords = {k: c_ for k, c_ in newx.coords.items() if k in cell_dims or k == dim}
        return DataArray(out.reshape(cell_shape + (out.shape[-1],)), cell_dims + (dim,), coords)
    out = hip.qm_adjust(x2.astype(dt, copy=False), yt.astype(dt, copy=False), xt.astype(dt, copy=False), gidx, method,
                        extrapolation, kind=None)
    coords = {k: c for k, c in newx.coords.items() if k in cell_dims or k == dim}
    return DataArray(out.reshape(cell_shape + (out.shape[-1],)), cell_dims + (dim,), coords)


@parse_group
def broadcast(grouped, x, *, group: str | Grouper = "time", interp: str = "nearest", sel=None):
    """Broadcast a grouped array back to the shape of `x` along time (utils.py:180-248): the value of each time step's
    group ("nearest") or the linear interpolation between the two neighbouring group centres on the cyclically extended
    table (a small host-side gather on the [.., group] table; the adjust kernels do the same on the device)."""
    grouped, x = as_dataarray(grouped), as_dataarray(x)
    if interp not in ("nearest", "linear", "cubic"):
        raise ValueError(f"interp must be 'nearest', 'linear' or 'cubic', got {interp!r}")
    if group.prop == "group":
        return grouped.squeeze("group") if "group" in grouped.dims else grouped
    gidx, G = group.group_index(x)
    ax = grouped.dims.index(group.prop)
    if interp == "cubic" and set(grouped.dims) & set(x.dims):  # utils.py:228-234
        warn("Broadcasting operations in multiple dimensions can only be done with linear and nearest-neighbor"
             " interpolation, not cubic. Using linear.", stacklevel=2)
        interp = "linear"
    if interp == "cubic":
        # xarray's .interp(method="cubic") = scipy's interp1d(kind="cubic") through the table wrapped with its cyclic rows
        from scipy.interpolate import interp1d

        i = np.asarray(group.get_index(x, interp=True).values, np.float64)
        c0 = 0 if group.prop == "season" else 1
        ext = np.concatenate([np.take(grouped.values, [G - 1], axis=ax), grouped.values, np.take(grouped.values, [0], axis=ax)], axis=ax)
        data = interp1d(np.arange(c0 - 1, c0 + G + 1, dtype=np.float64), ext.astype(np.float64), kind="cubic", axis=ax,
                        bounds_error=False)(i).astype(grouped.values.dtype)
    elif interp == "linear":
        i = np.asarray(group.get_index(x, interp=True).values, np.float64)
        c0 = 0 if group.prop == "season" else 1
        hi = np.ceil(i)
        lo = hi - 1
        shape = [1] * grouped.values.ndim
        shape[ax] = -1
        frac = (i - lo).reshape(shape)
        ylo = np.take(grouped.values, np.mod(lo - c0, G).astype(int), axis=ax)
        yhi = np.take(grouped.values, np.mod(hi - c0, G).astype(int), axis=ax)
        data = (yhi - ylo).astype(np.float64) * frac + ylo.astype(np.float64)  # scipy interp1d: slope * (x - x_lo) + y_lo
    else:
        data = np.take(grouped.values, gidx, axis=ax)
    dims = tuple(group.dim if d == group.prop else d for d in grouped.dims)
    coords = {k: c for k, c in grouped.coords.items() if k != group.prop}
    coords[group.dim] = x.coords[group.dim]
    return DataArray(data, dims, coords, grouped.attrs, grouped.name)


def rank(da, dim: str | list = "time", pct: bool = False, use_random_tiebreak: bool = False):
    """Rank data along a dimension (utils.py:573-638): average ties, NaN stay NaN; ``pct=True`` rescales to [0, 1].

    Computed by the HIP rank kernel (average ranks, or percent ranks).  Several dimensions are ranked together (the
    reference stacks them): the extra ones are laid end to end on the first.  ``use_random_tiebreak``: like the reference,
    U(0.1, 0.25) noise (numpy's global generator) is added to the average ranks, which are then ranked again.
    """
    da = as_dataarray(da)
    dims = dim if isinstance(dim, list) else [dim]
    main, extra = dims[0], [d for d in dims[1:] if d in da.dims]
    cell_dims = tuple(d for d in da.dims if d not in dims)
    t = da.transpose(*cell_dims, *extra, main)
    v = t.values.astype(np.result_type(t.values.dtype, np.float32), copy=False)
    cell_shape, pool_shape, T = v.shape[:len(cell_dims)], v.shape[len(cell_dims):-1], v.shape[-1]
    x2 = np.ascontiguousarray(v).reshape(int(np.prod(cell_shape, dtype=np.int64)), -1)  # [cell, pooled * T]
    if use_random_tiebreak:
        r = hip.rank_groups(x2, None, 1, 1, False, pct=False)
        noise = np.random.uniform(low=0.1, high=0.25, size=r.shape)
        x2 = hip.series_correction(r, noise, "+")
    pr = hip.rank_groups(x2, None, 1, 1, False, pct=bool(pct))
    coords = {k: c for k, c in da.coords.items()}
    out = DataArray(pr.reshape(cell_shape + pool_shape + (T,)), cell_dims + tuple(extra) + (main,), coords, da.attrs, da.name)
    return out.transpose(*da.dims)


__all__ = ["ADDITIVE", "MULTIPLICATIVE", "add_cyclic_bounds", "apply_correction", "broadcast", "equally_spaced_nodes",
           "get_correction", "interp_on_quantiles", "invert", "rank"]
