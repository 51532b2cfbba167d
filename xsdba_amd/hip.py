"""Array-level front-end of the HIP library: the functions the labelled (xarray-like) layer calls.

Every series argument is logically ``x[cell, time]``: a 2-D numpy array with ANY dense strides (so the
transposed view of a ``(time, cell)`` array is passed as is — the library reads time-major cubes through a
tiled layout kernel, no host copy) or a :class:`DeviceArray` (``.cell_time_strides`` tells the layout).
Results are numpy arrays for numpy inputs and :class:`DeviceArray` for device inputs.

Reference functions behind each call: see ``include/xsdba_hip.h``.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import EXTRAP, INTERP, Context, DeviceArray, check, default_context, dtype_code, load


def _vp(a):
    if a is None:
        return None
    if isinstance(a, DeviceArray):
        return C.c_void_p(a.ptr)
    return a.ctypes.data_as(C.c_void_p)


def _i64(v):
    return C.c_int64(int(v))


class _Series:
    """Normalised (pointer, n_cell, n_time, stride_t, stride_c) view of a logical [cell, time] block."""

    def __init__(self, x, dtype=None):
        if isinstance(x, DeviceArray):
            self.dev = True
            self.arr = x
            self.C, self.T = x.shape if getattr(x, "logical_shape", None) is None else x.logical_shape
            sc, st = getattr(x, "cell_time_strides", (self.T, 1))
            self.st, self.sc = int(st), int(sc)
            self.dtype = x.dtype
        else:
            x = np.asarray(x)
            if x.ndim != 2:
                raise ValueError("series must be 2-D [cell, time]")
            if dtype is not None and x.dtype != dtype:
                x = x.astype(dtype)
            self.C, self.T = x.shape
            sc, st = (s // x.itemsize for s in x.strides)
            cell_major = st == 1 and sc == self.T
            time_major = sc == 1 and st == self.C
            if not (cell_major or time_major) or self.C == 1 or self.T == 1:
                # sliced / broadcast / negative strides (or a degenerate axis): one dense host copy
                x = np.ascontiguousarray(x)
                sc, st = self.T, 1
            self.dev = False
            self.arr = x
            self.st, self.sc = int(st), int(sc)
            self.dtype = x.dtype

    @property
    def ptr(self):
        return _vp(self.arr)


def _gidx(gidx, T, G):
    if gidx is None:
        if G != 1:
            raise ValueError("group index required")
        return None
    g = np.ascontiguousarray(gidx, np.int32)
    if g.shape != (T,):
        raise ValueError(f"group_of_time must have shape ({T},), got {g.shape}")
    return g


def _out(ctx, dev, shape, dtype):
    return ctx.empty(shape, dtype) if dev else np.empty(shape, dtype)


def _table(tab, dtype, shape=None):
    if isinstance(tab, DeviceArray):
        return tab
    t = np.ascontiguousarray(tab, dtype)
    if shape is not None and t.shape != shape:
        raise ValueError(f"table shape {t.shape} != {shape}")
    return t


def quantile_groups(x, gidx, G, W, q, ctx: Context | None = None):
    """nbutils.quantile per (cell, group) -> [C, G, Q]"""
    ctx = ctx or default_context()
    s = _Series(x)
    q = np.ascontiguousarray(q, s.dtype)
    g = _gidx(gidx, s.T, G)
    out = _out(ctx, s.dev, (s.C, G, q.size), s.dtype)
    check(load().xsdba_hip_quantile_groups(ctx.handle, s.ptr, dtype_code(s.dtype), _i64(s.T), _i64(s.C), _i64(s.st),
                                           _i64(s.sc), _vp(g), C.c_int(G), C.c_int(W), _vp(q), C.c_int(q.size), _vp(out)),
          "quantile_groups")
    return out


def vecquantiles(x, rnk, ctx: Context | None = None):
    """nbutils.vecquantiles: out[c] = nanquantile(x[c, :], rnk[c]) -> [C]"""
    ctx = ctx or default_context()
    s = _Series(x)
    r = np.ascontiguousarray(rnk, s.dtype).reshape(-1)
    if r.size != s.C:
        raise ValueError(f"rnk must hold one quantile per row ({s.C}), got {r.size}")
    out = _out(ctx, s.dev, (s.C,), s.dtype)
    check(load().xsdba_hip_vecquantiles(ctx.handle, s.ptr, dtype_code(s.dtype), _i64(s.T), _i64(s.C), _i64(s.st), _i64(s.sc),
                                        _vp(r), _vp(out)), "vecquantiles")
    return out


def _train(fn, name, ref, hist, gidx, G, W, q, kind, with_scaling, ctx, out=None):
    ctx = ctx or default_context()
    r = _Series(ref)
    h = _Series(hist, r.dtype)
    if (h.C, h.T) != (r.C, r.T):
        raise ValueError("ref and hist must have the same shape")
    if (h.st, h.sc) != (r.st, r.sc) or h.dev != r.dev:
        if r.dev or h.dev:
            raise ValueError("ref and hist device arrays must share one layout")
        r = _Series(np.ascontiguousarray(r.arr))
        h = _Series(np.ascontiguousarray(h.arr))
    q = np.ascontiguousarray(q, r.dtype)
    g = _gidx(gidx, r.T, G)
    af = out[0] if out else _out(ctx, r.dev, (r.C, G, q.size), r.dtype)
    hq = out[1] if out else _out(ctx, r.dev, (r.C, G, q.size), r.dtype)
    args = [ctx.handle, r.ptr, h.ptr, dtype_code(r.dtype), _i64(r.T), _i64(r.C), _i64(r.st), _i64(r.sc), _vp(g),
            C.c_int(G), C.c_int(W), _vp(q), C.c_int(q.size), C.c_char(kind.encode()), _vp(af), _vp(hq)]
    if with_scaling:
        sc = out[2] if out else _out(ctx, r.dev, (r.C, G), r.dtype)
        check(fn(*args, _vp(sc)), name)
        return af, hq, sc
    check(fn(*args), name)
    return af, hq


def eqm_train(ref, hist, gidx, G, W, q, kind, ctx=None, out=None):
    """_adjustment.eqm_train -> (af, hist_q) each [C, G, Q]; ``out=(af, hist_q)`` reuses buffers"""
    return _train(load().xsdba_hip_eqm_train, "eqm_train", ref, hist, gidx, G, W, q, kind, False, ctx, out)


def dqm_train(ref, hist, gidx, G, W, q, kind, ctx=None, out=None):
    """_adjustment.dqm_train -> (af, hist_q, scaling); ``out=(af, hist_q, scaling)`` reuses buffers"""
    return _train(load().xsdba_hip_dqm_train, "dqm_train", ref, hist, gidx, G, W, q, kind, True, ctx, out)


def qm_adjust(sim, af, hist_q, gidx, interp="nearest", extrap="constant", kind="+", return_nodes=False, ctx=None,
              out_time_major=False, out=None):
    """_adjustment.qm_adjust -> scen [C, T]; ``kind=None`` returns the interpolated factors
    (utils.interp_on_quantiles)."""
    ctx = ctx or default_context()
    s = _Series(sim)
    G, Q = (af.shape[1], af.shape[2])
    af = _table(af, s.dtype, (s.C, G, Q))
    hist_q = _table(hist_q, s.dtype, (s.C, G, Q))
    g = _gidx(gidx, s.T, G)
    scen = out if out is not None else _out(ctx, s.dev, (s.T, s.C) if out_time_major else (s.C, s.T), s.dtype)
    ost, osc = (s.C, 1) if out_time_major else (1, s.T)
    node = np.empty((s.C, s.T), np.int32) if return_nodes else None
    k = C.c_char(kind.encode()) if kind else C.c_char(b"\0")
    check(load().xsdba_hip_qm_adjust(ctx.handle, s.ptr, dtype_code(s.dtype), _i64(s.T), _i64(s.C), _i64(s.st), _i64(s.sc),
                                     _vp(af), _vp(hist_q), C.c_int(G), C.c_int(Q), _vp(g), C.c_int(INTERP[interp]),
                                     C.c_int(EXTRAP[extrap]), k, _vp(scen), _i64(ost), _i64(osc), _vp(node)), "qm_adjust")
    if out_time_major and not s.dev:
        scen = scen.T
    return (scen, node) if return_nodes else scen


def qm_adjust_linear2d(sim, af, hist_q, group_coord, new_group, extrap="constant", kind="+", trend=None, lookup=None, ctx=None,
                       cubic=False):
    """qm_adjust with grouped interp="linear" (utils._interp_on_quantiles_2d through scipy.griddata, utils.py:380-400) -> scen
    [C, T]; ``cubic=True``: grouped interp="cubic" (griddata's Clough-Tocher interpolant, ``xsdba_hip_qm_adjust_cubic2d``).
    ``group_coord`` [G + 2]: coordinates of the table rows wrapped with the cyclic ones; ``new_group`` [T]: the fractional
    group coordinate of every time step; ``trend`` [C, T]: detrend before / retrend after the mapping.
    QDM form: ``hist_q`` a single row [Q] (the quantile nodes) and ``lookup`` [C, T] float64 (sim_q) to look up instead of sim."""
    ctx = ctx or default_context()
    s = _Series(sim)
    G, Q = (af.shape[1], af.shape[2])
    af = _table(af, s.dtype, (s.C, G, Q))
    row = tuple(hist_q.shape) == (Q,)
    hist_q = np.ascontiguousarray(hist_q, s.dtype) if row else _table(hist_q, s.dtype, (s.C, G, Q))
    gc = np.ascontiguousarray(group_coord, np.float64)
    ng = np.ascontiguousarray(new_group, np.float64)
    if gc.shape != (G + 2,) or ng.shape != (s.T,):
        raise ValueError(f"group_coord must have shape ({G + 2},) and new_group ({s.T},)")
    if trend is not None:
        trend = trend if isinstance(trend, DeviceArray) else np.ascontiguousarray(trend, s.dtype)
        if tuple(trend.shape) != (s.C, s.T):
            raise ValueError(f"trend must have shape {(s.C, s.T)}")
    if lookup is not None:
        lookup = lookup if isinstance(lookup, DeviceArray) else np.ascontiguousarray(lookup, np.float64)
        if tuple(lookup.shape) != (s.C, s.T):
            raise ValueError(f"lookup must have shape {(s.C, s.T)}")
    scen = _out(ctx, s.dev, (s.C, s.T), s.dtype)
    entry = load().xsdba_hip_qm_adjust_cubic2d if cubic else load().xsdba_hip_qm_adjust_linear2d
    check(entry(ctx.handle, s.ptr, dtype_code(s.dtype), _i64(s.T), _i64(s.C), _i64(s.st), _i64(s.sc),
                _vp(af), _vp(hist_q), C.c_int(G), C.c_int(Q), _vp(gc), _vp(ng), C.c_int(EXTRAP[extrap]),
                C.c_char(kind.encode() if kind else b"\0"), _vp(trend), _vp(lookup), C.c_int(int(row)),
                _vp(scen), _i64(1), _i64(s.T)),
          "qm_adjust_cubic2d" if cubic else "qm_adjust_linear2d")
    return scen


def dqm_adjust(sim, af, hist_q, scaling, gidx, W, time_num, degree=1, interp="nearest", extrap="constant", kind="+",
               ctx=None, out=None, trend_in=None, trend_kind=None, mult_skip_zeros=False, want_trend=True):
    """_adjustment.dqm_adjust -> (scen, trend) each [C, T] (``want_trend=False``: trend is None and, for a fitted line per
    group, never exists as an array: the mapping kernel evaluates the regression coefficients per element).
    ``trend_in`` ([C, T]): use this trend instead of fitting one
    (``degree`` and ``time_num`` are then ignored); ``trend_kind`` / ``mult_skip_zeros``: the detrending object's own kind
    and option (detrending.py:102-120), default: the kind of the adjustment."""
    ctx = ctx or default_context()
    s = _Series(sim)
    G, Q = (af.shape[1], af.shape[2])
    af = _table(af, s.dtype, (s.C, G, Q))
    hist_q = _table(hist_q, s.dtype, (s.C, G, Q))
    scaling = _table(scaling, s.dtype, (s.C, G))
    g = _gidx(gidx, s.T, G)
    scen = out[0] if out else _out(ctx, s.dev, (s.C, s.T), s.dtype)
    if trend_in is not None:
        trend = trend_in if isinstance(trend_in, DeviceArray) else np.ascontiguousarray(trend_in, s.dtype)
        if tuple(trend.shape) != (s.C, s.T):
            raise ValueError(f"trend_in must have shape {(s.C, s.T)}")
        tk = trend_kind or kind
        # XSDBA_TREND_GIVEN / _ADD / _MUL / _MUL_SKIP_ZEROS
        tn, degree = None, (-1 if tk == kind and not mult_skip_zeros else (-2 if tk == "+" else (-4 if mult_skip_zeros else -3)))
    else:
        tn = time_num if isinstance(time_num, DeviceArray) else np.ascontiguousarray(time_num, np.float64)
        trend = (out[1] if out else _out(ctx, s.dev, (s.C, s.T), s.dtype)) if want_trend else None
    check(load().xsdba_hip_dqm_adjust(ctx.handle, s.ptr, dtype_code(s.dtype), _i64(s.T), _i64(s.C), _i64(s.st), _i64(s.sc),
                                      _vp(af), _vp(hist_q), _vp(scaling), C.c_int(G), C.c_int(Q), _vp(g), C.c_int(W),
                                      _vp(tn), C.c_int(degree), C.c_int(INTERP[interp]), C.c_int(EXTRAP[extrap]),
                                      C.c_char(kind.encode()), _vp(scen), _vp(trend), _i64(1), _i64(s.T)), "dqm_adjust")
    return scen, trend


def qdm_adjust(sim, af, quantiles, gidx, W, rank_window=False, interp="nearest", extrap="constant", kind="+",
               want_sim_q=True, ctx=None, out=None):
    """_adjustment.qdm_adjust -> (scen, sim_q)"""
    ctx = ctx or default_context()
    s = _Series(sim)
    G, Q = (af.shape[1], af.shape[2])
    af = _table(af, s.dtype, (s.C, G, Q))
    quantiles = np.ascontiguousarray(quantiles, s.dtype)
    g = _gidx(gidx, s.T, G)
    scen = out if out is not None else _out(ctx, s.dev, (s.C, s.T), s.dtype)
    sim_q = _out(ctx, s.dev, (s.C, s.T), np.float64) if want_sim_q else None
    check(load().xsdba_hip_qdm_adjust(ctx.handle, s.ptr, dtype_code(s.dtype), _i64(s.T), _i64(s.C), _i64(s.st), _i64(s.sc),
                                      _vp(af), _vp(quantiles), C.c_int(G), C.c_int(Q), _vp(g), C.c_int(W),
                                      C.c_int(int(bool(rank_window))), C.c_int(INTERP[interp]), C.c_int(EXTRAP[extrap]),
                                      C.c_char(kind.encode()), _vp(scen), _vp(sim_q), _i64(1), _i64(s.T)), "qdm_adjust")
    return scen, sim_q


def tail_mask(sim, scen, hist_q_raw, gidx, max_tail_factor, ctx=None):
    """the max_tail_factor mask (_adjustment.py:648-658, 672-673), in place on ``scen`` [C, T]: values of ``sim`` above
    ``max_tail_factor`` times the last raw hist quantile of their group keep the input value"""
    ctx = ctx or default_context()
    s = _Series(sim)
    G, Q = hist_q_raw.shape[1], hist_q_raw.shape[2]
    hq = _table(hist_q_raw, s.dtype, (s.C, G, Q))
    g = _gidx(gidx, s.T, G)
    if not s.dev and not (isinstance(scen, np.ndarray) and scen.dtype == s.dtype and scen.flags.c_contiguous and scen.shape == (s.C, s.T)):
        raise ValueError("scen must be a C-contiguous [cell, time] array of the series dtype")
    check(load().xsdba_hip_tail_mask(ctx.handle, s.ptr, dtype_code(s.dtype), _i64(s.T), _i64(s.C), _i64(s.st), _i64(s.sc),
                                     _vp(hq), C.c_int(G), C.c_int(Q), _vp(g), C.c_double(float(max_tail_factor)), _vp(scen),
                                     _i64(1), _i64(s.T)), "tail_mask")
    return scen


def jitter(x, lower=None, minimum=None, upper=None, maximum=None, seed=0, first_cell=0, ctx=None):
    """processing.jitter (processing.py:178-257) on a [cell, time] block; thresholds are plain numbers in the units of x"""
    ctx = ctx or default_context()
    s = _Series(x)
    out = _out(ctx, s.dev, (s.C, s.T), s.dtype)
    nan = float("nan")
    f = lambda v: nan if v is None else float(v)  # noqa: E731
    check(load().xsdba_hip_jitter(ctx.handle, s.ptr, dtype_code(s.dtype), _i64(s.T), _i64(s.C), _i64(s.st), _i64(s.sc),
                                  C.c_double(f(lower)), C.c_double(f(minimum)), C.c_double(f(upper)), C.c_double(f(maximum)),
                                  C.c_uint64(int(seed) & (2**64 - 1)), _i64(first_cell), _vp(out), _i64(1), _i64(s.T)), "jitter")
    return out


def adapt_freq(sim, ref, gidx, G, W, thresh, seed=0, first_cell=0, P0_ref=None, P0_hist=None, pth=None, ctx=None):
    """_processing._adapt_freq (_processing.py:20-142) on every (cell, group) block.
    Training (``ref`` given): returns (sim_ad, pth, dP0, P0_ref, P0_hist).  Adjusting (``ref=None``; P0_ref, P0_hist, pth
    from the training, groups without window): returns the same tuple with the given tables passed through."""
    ctx = ctx or default_context()
    s = _Series(sim)
    g = _gidx(gidx, s.T, G)
    train = ref is not None
    if train:
        r = _Series(ref)
        if (r.st, r.sc) != (s.st, s.sc) and isinstance(sim, np.ndarray) and isinstance(ref, np.ndarray):
            # one pair of strides serves both series: bring host arrays to the same dense [cell, time] layout
            sim, ref = np.ascontiguousarray(sim), np.ascontiguousarray(ref)
            s, r = _Series(sim), _Series(ref)
        if (r.C, r.T, r.dtype, r.st, r.sc) != (s.C, s.T, s.dtype, s.st, s.sc):
            raise ValueError("ref and sim must share shape, dtype and layout")
        P0_ref, P0_hist = np.empty((s.C, G), np.float64), np.empty((s.C, G), np.float64)
        pth = np.empty((s.C, G), s.dtype)
    else:
        if W != 1:
            raise ValueError("adapt_freq on the adjust side works on the groups without window")
        P0_ref, P0_hist = np.ascontiguousarray(P0_ref, np.float64), np.ascontiguousarray(P0_hist, np.float64)
        pth = np.ascontiguousarray(pth, s.dtype)
        if P0_ref.shape != (s.C, G) or P0_hist.shape != (s.C, G) or pth.shape != (s.C, G):
            raise ValueError("P0_ref, P0_hist, pth must be [cell, group]")
    dP0 = np.empty((s.C, G), np.float64)
    sim_ad = _out(ctx, s.dev, (s.C, s.T), s.dtype)
    check(load().xsdba_hip_adapt_freq(ctx.handle, s.ptr, r.ptr if train else None, dtype_code(s.dtype), _i64(s.T), _i64(s.C),
                                      _i64(s.st), _i64(s.sc), _vp(g), C.c_int(G), C.c_int(W), C.c_double(float(thresh)),
                                      C.c_uint64(int(seed) & (2**64 - 1)), _i64(first_cell), _vp(P0_ref), _vp(P0_hist), _vp(pth),
                                      _vp(dP0), _vp(sim_ad), _i64(1), _i64(s.T)), "adapt_freq")
    return sim_ad, pth, dP0, P0_ref, P0_hist


def group_means(x, gidx, G, W, ctx=None):
    """NaN-skipping mean of every (cell, group) block, dims [time, window] pooled -> [C, G] (scaling_train,
    _adjustment.py:953-956)"""
    ctx = ctx or default_context()
    s = _Series(x)
    g = _gidx(gidx, s.T, G)
    out = np.empty((s.C, G), s.dtype)
    check(load().xsdba_hip_group_means(ctx.handle, s.ptr, dtype_code(s.dtype), _i64(s.T), _i64(s.C), _i64(s.st), _i64(s.sc),
                                       _vp(g), C.c_int(G), C.c_int(W), _vp(out)), "group_means")
    return out


def _lin(g_hi, frac, T):
    if g_hi is None and frac is None:
        return None, None
    g_hi, frac = np.ascontiguousarray(g_hi, np.int32), np.ascontiguousarray(frac, np.float64)
    if g_hi.shape != (T,) or frac.shape != (T,):
        raise ValueError("g_hi and frac must have one entry per time step")
    return g_hi, frac


def scaling_adjust(sim, af, gidx, kind="+", g_hi=None, frac=None, ctx=None):
    """scaling_adjust (_adjustment.py:961-974) -> scen [C, T]; nearest-group factors, or (``g_hi``, ``frac`` given, ``gidx`` =
    lower neighbour) their linear interpolation between neighbouring groups"""
    ctx = ctx or default_context()
    s = _Series(sim)
    G = af.shape[1]
    af = _table(af, s.dtype, (s.C, G))
    g = _gidx(gidx, s.T, G)
    g_hi, frac = _lin(g_hi, frac, s.T)
    scen = _out(ctx, s.dev, (s.C, s.T), s.dtype)
    check(load().xsdba_hip_scaling_adjust(ctx.handle, s.ptr, dtype_code(s.dtype), _i64(s.T), _i64(s.C), _i64(s.st), _i64(s.sc),
                                          _vp(af), _vp(g), C.c_int(G), _vp(g_hi), _vp(frac), C.c_char(kind.encode()), _vp(scen),
                                          _i64(1), _i64(s.T)), "scaling_adjust")
    return scen


def loci_train(ref, hist, gidx, G, W, thresh, ctx=None):
    """loci_train (_adjustment.py:889-919) -> (af, hist_thresh) each [C, G]"""
    ctx = ctx or default_context()
    r, h = _Series(ref), _Series(hist)
    if (r.st, r.sc) != (h.st, h.sc) and isinstance(ref, np.ndarray) and isinstance(hist, np.ndarray):
        ref, hist = np.ascontiguousarray(ref), np.ascontiguousarray(hist)
        r, h = _Series(ref), _Series(hist)
    if (r.C, r.T, r.dtype, r.st, r.sc) != (h.C, h.T, h.dtype, h.st, h.sc):
        raise ValueError("ref and hist must share shape, dtype and layout")
    g = _gidx(gidx, r.T, G)
    af, sth = np.empty((r.C, G), r.dtype), np.empty((r.C, G), r.dtype)
    check(load().xsdba_hip_loci_train(ctx.handle, r.ptr, h.ptr, dtype_code(r.dtype), _i64(r.T), _i64(r.C), _i64(r.st), _i64(r.sc),
                                      _vp(g), C.c_int(G), C.c_int(W), C.c_double(float(thresh)), _vp(af), _vp(sth)), "loci_train")
    return af, sth


def loci_adjust(sim, af, hist_thresh, gidx, thresh, g_hi=None, frac=None, ctx=None):
    """loci_adjust (_adjustment.py:922-939) -> scen [C, T]; nearest-group or linearly interpolated factors / thresholds"""
    ctx = ctx or default_context()
    s = _Series(sim)
    G = af.shape[1]
    af, sth = _table(af, s.dtype, (s.C, G)), _table(hist_thresh, s.dtype, (s.C, G))
    g = _gidx(gidx, s.T, G)
    g_hi, frac = _lin(g_hi, frac, s.T)
    scen = _out(ctx, s.dev, (s.C, s.T), s.dtype)
    check(load().xsdba_hip_loci_adjust(ctx.handle, s.ptr, dtype_code(s.dtype), _i64(s.T), _i64(s.C), _i64(s.st), _i64(s.sc),
                                       _vp(af), _vp(sth), _vp(g), C.c_int(G), _vp(g_hi), _vp(frac), C.c_double(float(thresh)),
                                       _vp(scen), _i64(1), _i64(s.T)), "loci_adjust")
    return scen


def series_correction(x, f, kind="+", inverted=False, skip_zeros=False, ctx=None):
    """utils.apply_correction(x, f or invert(f), kind) for series x [C, T] and f [C, T] or [T] (broadcast) -> [C, T]"""
    ctx = ctx or default_context()
    s = _Series(x)
    f = np.ascontiguousarray(f, s.dtype)
    if f.shape == (s.T,):
        fst, fsc = 1, 0
    elif f.shape == (s.C, s.T):
        fst, fsc = 1, s.T
    else:
        raise ValueError(f"f must have shape {(s.C, s.T)} or {(s.T,)}")
    out = _out(ctx, s.dev, (s.C, s.T), s.dtype)
    check(load().xsdba_hip_series_correction(ctx.handle, s.ptr, dtype_code(s.dtype), _i64(s.T), _i64(s.C), _i64(s.st), _i64(s.sc),
                                             _vp(f), _i64(fst), _i64(fsc), C.c_char(kind.encode()), C.c_int(int(bool(inverted))),
                                             C.c_int(int(bool(skip_zeros))), _vp(out), _i64(1), _i64(s.T)), "series_correction")
    return out


def rolling_mean(x, window, min_periods=None, ctx=None):
    """x.rolling(time=window, center=True, min_periods=min_periods).mean() -> [C, T] (min_periods defaults to window)"""
    ctx = ctx or default_context()
    s = _Series(x)
    out = _out(ctx, s.dev, (s.C, s.T), s.dtype)
    mp = int(window if min_periods is None else min_periods)
    check(load().xsdba_hip_rolling_mean(ctx.handle, s.ptr, dtype_code(s.dtype), _i64(s.T), _i64(s.C), _i64(s.st), _i64(s.sc),
                                        C.c_int(int(window)), C.c_int(mp), _vp(out), _i64(1), _i64(s.T)), "rolling_mean")
    return out


def rank_groups(x, gidx, G, W, rank_window=False, pct=True, ctx=None):
    """utils.rank(pct) through Grouper.apply -> float64 [C, T] (average ranks, or percent ranks rescaled to [0, 1])"""
    ctx = ctx or default_context()
    s = _Series(x)
    g = _gidx(gidx, s.T, G)
    out = _out(ctx, s.dev, (s.C, s.T), np.float64)
    check(load().xsdba_hip_rank_groups(ctx.handle, s.ptr, dtype_code(s.dtype), _i64(s.T), _i64(s.C), _i64(s.st), _i64(s.sc),
                                       _vp(g), C.c_int(G), C.c_int(W), C.c_int(int(bool(rank_window)) | (0 if pct else 2)), _vp(out),
                                       _i64(1), _i64(s.T)), "rank_groups")
    return out


def polydetrend_trend(x, gidx, G, W, time_num, degree=1, preserve_mean=False, kind="+", ctx=None):
    """PolyDetrend(degree, preserve_mean, kind).fit(x).ds.trend -> [C, T]"""
    ctx = ctx or default_context()
    s = _Series(x)
    g = _gidx(gidx, s.T, G)
    tn = np.ascontiguousarray(time_num, np.float64)
    out = _out(ctx, s.dev, (s.C, s.T), s.dtype)
    check(load().xsdba_hip_polydetrend_fit(ctx.handle, s.ptr, dtype_code(s.dtype), _i64(s.T), _i64(s.C), _i64(s.st),
                                           _i64(s.sc), _vp(g), C.c_int(G), C.c_int(W), _vp(tn), C.c_int(degree),
                                           C.c_int(int(bool(preserve_mean))), C.c_char(kind.encode()), _vp(out),
                                           _i64(1), _i64(s.T)), "polydetrend_fit")
    return out


__all__ = ["vecquantiles", "quantile_groups", "eqm_train", "dqm_train", "qm_adjust", "qm_adjust_linear2d", "dqm_adjust", "qdm_adjust", "tail_mask", "jitter", "adapt_freq", "group_means", "scaling_adjust", "series_correction", "rolling_mean", "loci_train", "loci_adjust", "rank_groups",
           "polydetrend_trend", "Context", "DeviceArray", "default_context"]
