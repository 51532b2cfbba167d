"""ctypes front-end of the CPU oracle (``oracle/xsdba_oracle.c``).

TEST INFRASTRUCTURE ONLY: imported by ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` — never by ``xsdba_amd`` (the product fails loudly when its
HIP library is missing instead of falling back to this).

All functions take cell-major numpy arrays ``x[C, T]`` (any strides), 0-based ``gidx[T]`` (int32
group of the window centre), and return new arrays; see the C file for the reference citations.

The functions at the end of this file (``interp_on_quantiles_2d``, ``interp_on_quantiles_1d``, ``polyfit_trend``) are plain
numpy / scipy: they call the third-party routine the reference itself calls (scipy griddata / interp1d, numpy lstsq)
composed as the cited reference lines do — the checker for the rows whose arithmetic lives in those libraries.
"""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB = None

INTERP = {"nearest": 0, "linear": 1}
EXTRAP = {"constant": 0, "nan": 1}


def build(force: bool = False) -> Path:
    so = _HERE / "_build" / "libxsdba_oracle.so"
    src = [_HERE / "xsdba_oracle.c", _HERE / "oracle_impl.inc"]
    if force or not so.exists() or any(s.stat().st_mtime > so.stat().st_mtime for s in src):
        subprocess.run(["make", "-C", str(_HERE)], check=True, capture_output=True)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(str(build()))
        _LIB.xsdba_oracle_num_threads.restype = C.c_int
    return _LIB


def _dt(a):
    if a.dtype == np.float32:
        return 0
    if a.dtype == np.float64:
        return 1
    raise TypeError(f"unsupported dtype {a.dtype}")


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _strides(a):
    """element strides (stride_t, stride_c) of a [C, T] array"""
    return a.strides[1] // a.itemsize, a.strides[0] // a.itemsize


def _i64(v):
    return C.c_int64(int(v))


def _check(rc, what):
    if rc != 0:
        raise RuntimeError(f"oracle {what} failed with code {rc}")


def num_threads() -> int:
    return lib().xsdba_oracle_num_threads()


def set_num_threads(n: int) -> None:
    lib().xsdba_oracle_set_num_threads(C.c_int(n))


def quantile_groups(x, gidx, G, W, q):
    x = np.asarray(x)
    gidx = np.ascontiguousarray(gidx, np.int32)
    q = np.ascontiguousarray(q, x.dtype)
    Cn, T = x.shape
    out = np.empty((Cn, G, q.size), x.dtype)
    st, sc = _strides(x)
    _check(lib().xsdba_oracle_quantile_groups(_p(x), _dt(x), _i64(T), _i64(Cn), _i64(st), _i64(sc), _p(gidx),
                                              C.c_int(G), C.c_int(W), _p(q), C.c_int(q.size), _p(out)), "quantile_groups")
    return out


def eqm_train(ref, hist, gidx, G, W, q, kind):
    ref = np.asarray(ref)
    hist = np.ascontiguousarray(hist, ref.dtype) if hist.strides != ref.strides else hist
    gidx = np.ascontiguousarray(gidx, np.int32)
    q = np.ascontiguousarray(q, ref.dtype)
    Cn, T = ref.shape
    af = np.empty((Cn, G, q.size), ref.dtype)
    hq = np.empty_like(af)
    if hist.strides != ref.strides:
        ref = np.ascontiguousarray(ref)
    st, sc = _strides(ref)
    _check(lib().xsdba_oracle_eqm_train(_p(ref), _p(hist), _dt(ref), _i64(T), _i64(Cn), _i64(st), _i64(sc), _p(gidx),
                                        C.c_int(G), C.c_int(W), _p(q), C.c_int(q.size), C.c_char(kind.encode()),
                                        _p(af), _p(hq)), "eqm_train")
    return af, hq


def dqm_train(ref, hist, gidx, G, W, q, kind):
    ref = np.ascontiguousarray(ref)
    hist = np.ascontiguousarray(hist, ref.dtype)
    gidx = np.ascontiguousarray(gidx, np.int32)
    q = np.ascontiguousarray(q, ref.dtype)
    Cn, T = ref.shape
    af = np.empty((Cn, G, q.size), ref.dtype)
    hq = np.empty_like(af)
    sc_ = np.empty((Cn, G), ref.dtype)
    st, sc = _strides(ref)
    _check(lib().xsdba_oracle_dqm_train(_p(ref), _p(hist), _dt(ref), _i64(T), _i64(Cn), _i64(st), _i64(sc), _p(gidx),
                                        C.c_int(G), C.c_int(W), _p(q), C.c_int(q.size), C.c_char(kind.encode()),
                                        _p(af), _p(hq), _p(sc_)), "dqm_train")
    return af, hq, sc_


def qm_adjust(sim, af, hist_q, gidx, interp="nearest", extrap="constant", kind="+", return_nodes=False):
    """``kind=None`` returns the interpolated factors themselves (utils.interp_on_quantiles)."""
    sim = np.asarray(sim)
    af = np.ascontiguousarray(af, sim.dtype)
    hist_q = np.ascontiguousarray(hist_q, sim.dtype)
    Cn, T = sim.shape
    G, Q = af.shape[1:]
    gidx = np.ascontiguousarray(gidx if gidx is not None else np.zeros(T), np.int32)
    scen = np.empty((Cn, T), sim.dtype)
    node = np.empty((Cn, T), np.int32) if return_nodes else None
    st, sc = _strides(sim)
    k = C.c_char(kind.encode()) if kind else C.c_char(b"\0")
    _check(lib().xsdba_oracle_qm_adjust(_p(sim), _dt(sim), _i64(T), _i64(Cn), _i64(st), _i64(sc), _p(af), _p(hist_q),
                                        C.c_int(G), C.c_int(Q), _p(gidx), C.c_int(INTERP[interp]),
                                        C.c_int(EXTRAP[extrap]), k, _p(scen), _i64(1), _i64(T), _p(node)), "qm_adjust")
    return (scen, node) if return_nodes else scen


def dqm_adjust(sim, af, hist_q, scaling, gidx, W, time_num, degree=1, interp="nearest", extrap="constant", kind="+"):
    sim = np.asarray(sim)
    af = np.ascontiguousarray(af, sim.dtype)
    hist_q = np.ascontiguousarray(hist_q, sim.dtype)
    scaling = np.ascontiguousarray(scaling, sim.dtype)
    Cn, T = sim.shape
    G, Q = af.shape[1:]
    gidx = np.ascontiguousarray(gidx, np.int32)
    tn = np.ascontiguousarray(time_num, np.float64)
    scen = np.empty((Cn, T), sim.dtype)
    trend = np.empty((Cn, T), sim.dtype)
    st, sc = _strides(sim)
    _check(lib().xsdba_oracle_dqm_adjust(_p(sim), _dt(sim), _i64(T), _i64(Cn), _i64(st), _i64(sc), _p(af), _p(hist_q),
                                         _p(scaling), C.c_int(G), C.c_int(Q), _p(gidx), C.c_int(W), _p(tn),
                                         C.c_int(degree), C.c_int(INTERP[interp]), C.c_int(EXTRAP[extrap]),
                                         C.c_char(kind.encode()), _p(scen), _p(trend), _i64(1), _i64(T)), "dqm_adjust")
    return scen, trend


def qdm_adjust(sim, af, quantiles, gidx, W, rank_window=False, interp="nearest", extrap="constant", kind="+"):
    sim = np.asarray(sim)
    af = np.ascontiguousarray(af, sim.dtype)
    quantiles = np.ascontiguousarray(quantiles, sim.dtype)
    Cn, T = sim.shape
    G, Q = af.shape[1:]
    gidx = np.ascontiguousarray(gidx, np.int32)
    scen = np.empty((Cn, T), sim.dtype)
    sim_q = np.empty((Cn, T), np.float64)
    st, sc = _strides(sim)
    _check(lib().xsdba_oracle_qdm_adjust(_p(sim), _dt(sim), _i64(T), _i64(Cn), _i64(st), _i64(sc), _p(af),
                                         _p(quantiles), C.c_int(G), C.c_int(Q), _p(gidx), C.c_int(W),
                                         C.c_int(int(bool(rank_window))), C.c_int(INTERP[interp]),
                                         C.c_int(EXTRAP[extrap]), C.c_char(kind.encode()), _p(scen), _p(sim_q),
                                         _i64(1), _i64(T)), "qdm_adjust")
    return scen, sim_q


def tail_mask(sim, scen, hist_q_raw, gidx, max_tail_factor):
    """scen with the max_tail_factor mask applied (a copy)"""
    sim = np.asarray(sim)
    Cn, T = sim.shape
    out = np.array(scen, dtype=sim.dtype, order="C", copy=True)
    hq = np.ascontiguousarray(hist_q_raw, sim.dtype)
    G, Q = hq.shape[1], hq.shape[2]
    g = np.zeros(T, np.int32) if gidx is None else np.ascontiguousarray(gidx, np.int32)
    st, sc = _strides(sim)
    _check(lib().xsdba_oracle_tail_mask(_p(sim), _dt(sim), _i64(T), _i64(Cn), _i64(st), _i64(sc), _p(hq), C.c_int(G), C.c_int(Q),
                                        _p(g), C.c_double(float(max_tail_factor)), _p(out), _i64(1), _i64(T)), "tail_mask")
    return out


def jitter(x, lower=None, minimum=None, upper=None, maximum=None, seed=0, first_cell=0):
    x = np.ascontiguousarray(x)
    Cn, T = x.shape
    out = np.empty_like(x)
    nan = float("nan")
    f = lambda v: nan if v is None else float(v)  # noqa: E731
    _check(lib().xsdba_oracle_jitter(_p(x), _dt(x), _i64(T), _i64(Cn), _i64(1), _i64(T), C.c_double(f(lower)), C.c_double(f(minimum)),
                                     C.c_double(f(upper)), C.c_double(f(maximum)), C.c_uint64(int(seed)), _i64(first_cell), _p(out),
                                     _i64(1), _i64(T)), "jitter")
    return out


def adapt_freq(sim, ref, gidx, G, W, thresh, seed=0, first_cell=0, P0_ref=None, P0_hist=None, pth=None):
    """returns (sim_ad, pth, dP0, P0_ref, P0_hist)"""
    sim = np.ascontiguousarray(sim)
    Cn, T = sim.shape
    gidx = np.ascontiguousarray(gidx if gidx is not None else np.zeros(T, np.int32), np.int32)
    if ref is not None:
        ref = np.ascontiguousarray(ref, sim.dtype)
        P0_ref, P0_hist, pth = np.empty((Cn, G)), np.empty((Cn, G)), np.empty((Cn, G), sim.dtype)
    else:
        P0_ref, P0_hist = np.ascontiguousarray(P0_ref, np.float64), np.ascontiguousarray(P0_hist, np.float64)
        pth = np.ascontiguousarray(pth, sim.dtype)
    dP0 = np.empty((Cn, G))
    out = np.empty_like(sim)
    _check(lib().xsdba_oracle_adapt_freq(_p(sim), _p(ref) if ref is not None else None, _dt(sim), _i64(T), _i64(Cn), _i64(1), _i64(T),
                                         _p(gidx), C.c_int(G), C.c_int(W), C.c_double(float(thresh)), C.c_uint64(int(seed)),
                                         _i64(first_cell), _p(P0_ref), _p(P0_hist), _p(pth), _p(dP0), _p(out), _i64(1), _i64(T)),
           "adapt_freq")
    return out, pth, dP0, P0_ref, P0_hist


def group_means(x, gidx, G, W):
    x = np.ascontiguousarray(x)
    Cn, T = x.shape
    gidx = np.ascontiguousarray(gidx if gidx is not None else np.zeros(T, np.int32), np.int32)
    out = np.empty((Cn, G), x.dtype)
    _check(lib().xsdba_oracle_group_means(_p(x), _dt(x), _i64(T), _i64(Cn), _i64(1), _i64(T), _p(gidx), C.c_int(G), C.c_int(W), _p(out)),
           "group_means")
    return out


def _lin(g_hi, frac):
    if g_hi is None:
        return None, None
    return np.ascontiguousarray(g_hi, np.int32), np.ascontiguousarray(frac, np.float64)


def scaling_adjust(sim, af, gidx, kind="+", g_hi=None, frac=None):
    sim = np.ascontiguousarray(sim)
    Cn, T = sim.shape
    af = np.ascontiguousarray(af, sim.dtype)
    G = af.shape[1]
    gidx = np.ascontiguousarray(gidx if gidx is not None else np.zeros(T, np.int32), np.int32)
    g_hi, frac = _lin(g_hi, frac)
    out = np.empty_like(sim)
    _check(lib().xsdba_oracle_scaling_adjust(_p(sim), _dt(sim), _i64(T), _i64(Cn), _i64(1), _i64(T), _p(af), _p(gidx), C.c_int(G),
                                             _p(g_hi), _p(frac), C.c_char(kind.encode()), _p(out), _i64(1), _i64(T)), "scaling_adjust")
    return out


def loci_train(ref, hist, gidx, G, W, thresh):
    """-> (af, hist_thresh) each [C, G]"""
    ref = np.ascontiguousarray(ref)
    hist = np.ascontiguousarray(hist, ref.dtype)
    Cn, T = ref.shape
    gidx = np.ascontiguousarray(gidx if gidx is not None else np.zeros(T, np.int32), np.int32)
    af, sth = np.empty((Cn, G), ref.dtype), np.empty((Cn, G), ref.dtype)
    _check(lib().xsdba_oracle_loci_train(_p(ref), _p(hist), _dt(ref), _i64(T), _i64(Cn), _i64(1), _i64(T), _p(gidx), C.c_int(G),
                                         C.c_int(W), C.c_double(float(thresh)), _p(af), _p(sth)), "loci_train")
    return af, sth


def loci_adjust(sim, af, hist_thresh, gidx, thresh, g_hi=None, frac=None):
    sim = np.ascontiguousarray(sim)
    Cn, T = sim.shape
    af, sth = np.ascontiguousarray(af, sim.dtype), np.ascontiguousarray(hist_thresh, sim.dtype)
    G = af.shape[1]
    gidx = np.ascontiguousarray(gidx if gidx is not None else np.zeros(T, np.int32), np.int32)
    out = np.empty_like(sim)
    g_hi, frac = _lin(g_hi, frac)
    _check(lib().xsdba_oracle_loci_adjust(_p(sim), _dt(sim), _i64(T), _i64(Cn), _i64(1), _i64(T), _p(af), _p(sth), _p(gidx), C.c_int(G),
                                          _p(g_hi), _p(frac), C.c_double(float(thresh)), _p(out), _i64(1), _i64(T)), "loci_adjust")
    return out


def rank_groups(x, gidx, G, W, rank_window=False):
    x = np.asarray(x)
    Cn, T = x.shape
    gidx = np.ascontiguousarray(gidx, np.int32)
    out = np.empty((Cn, T), np.float64)
    st, sc = _strides(x)
    _check(lib().xsdba_oracle_rank_groups(_p(x), _dt(x), _i64(T), _i64(Cn), _i64(st), _i64(sc), _p(gidx), C.c_int(G),
                                          C.c_int(W), C.c_int(int(bool(rank_window))), _p(out), _i64(1), _i64(T)),
           "rank_groups")
    return out


def polydetrend_trend(x, gidx, G, W, time_num, degree=1):
    x = np.asarray(x)
    Cn, T = x.shape
    gidx = np.ascontiguousarray(gidx, np.int32)
    tn = np.ascontiguousarray(time_num, np.float64)
    out = np.empty((Cn, T), x.dtype)
    st, sc = _strides(x)
    _check(lib().xsdba_oracle_polydetrend_trend(_p(x), _dt(x), _i64(T), _i64(Cn), _i64(st), _i64(sc), _p(gidx),
                                                C.c_int(G), C.c_int(W), _p(tn), C.c_int(degree), _p(out), _i64(1),
                                                _i64(T)), "polydetrend_trend")
    return out


def vecquantiles(x, rnk):
    x = np.asarray(x)
    Cn, T = x.shape
    r = np.ascontiguousarray(rnk, x.dtype).reshape(-1)
    out = np.empty(Cn, x.dtype)
    st, sc = _strides(x)
    _check(lib().xsdba_oracle_vecquantiles(_p(x), _dt(x), _i64(T), _i64(Cn), _i64(st), _i64(sc), _p(r), _p(out)),
           "vecquantiles")
    return out


def interp_on_quantiles_2d(newx, newg, oldx, oldy, oldg, method="linear", extrap="constant"):
    """utils._interp_on_quantiles_2d (utils.py:380-400) + nbutils._extrapolate_on_quantiles (nbutils.py:397-416) for ONE
    cell, with scipy.interpolate.griddata itself (the third-party routine the reference calls, scipy >= 1.11; 1.15 here):
    oldx, oldy, oldg are the [G + 2, Q] table rows wrapped with the cyclic ones and their group coordinates, newx / newg the
    series and the (fractional) group coordinate of every time step.  Output in the dtype of oldy, like the reference."""
    from scipy.interpolate import griddata

    newx, newg = np.asarray(newx), np.asarray(newg, np.float64)
    oldx, oldy, oldg = np.asarray(oldx), np.asarray(oldy), np.asarray(oldg, np.float64)
    mask_new = np.isnan(newx) | np.isnan(newg)
    mask_old = np.isnan(oldy) | np.isnan(oldx) | np.isnan(oldg)
    out = np.full(newx.shape, np.nan, oldy.dtype)
    if mask_new.all() or mask_old.all():
        return out
    out[~mask_new] = griddata((oldx[~mask_old], oldg[~mask_old]), oldy[~mask_old], (newx[~mask_new], newg[~mask_new]), method=method)
    if method == "nearest" or extrap != "nan":
        def first_last(arr):
            fl = np.full((arr.shape[0], 2), np.nan)
            for i, row in enumerate(arr):
                ok = np.nonzero(~np.isnan(row))[0]
                if ok.size:
                    fl[i] = row[ok[[0, -1]]]
            return fl

        bnds, xp = first_last(oldx), oldg[:, 0]
        toolow = newx < np.interp(newg, xp, bnds[:, 0])
        toohigh = newx > np.interp(newg, xp, bnds[:, 1])
        if extrap == "constant":
            cst = first_last(oldy)
            out[toolow] = np.interp(newg, xp, cst[:, 0])[toolow]
            out[toohigh] = np.interp(newg, xp, cst[:, 1])[toohigh]
        else:
            out[toolow] = np.nan
            out[toohigh] = np.nan
    return out


def polyfit_trend(x, gidx, G, W, time_num, degree, preserve_mean=False, kind="+"):
    """detrending._polydetrend_get_trend (detrending.py:196-208) for any degree, restated with numpy exactly as
    xarray.DataArray.polyfit / xr.polyval do it (xarray/core/dataset.py polyfit: Vandermonde matrix of the float64 time axis,
    columns scaled by their 2-norm, numpy.linalg.lstsq with rcond = n * eps on the non-NaN rows, coefficients / scale;
    polyval: Horner on the same axis): x [C, T] -> trend [C, T] in x's dtype.  Window means: NaN-skipping mean of the
    centred window of every time step (rolling.construct + mean(skipna), base.py:261-262)."""
    x = np.asarray(x)
    gidx = np.asarray(gidx)
    tn = np.asarray(time_num, np.float64)
    C, T = x.shape
    h = W // 2
    out = np.full((C, T), np.nan, np.float64)
    for c in range(C):
        if W > 1:
            pad = np.concatenate([np.full(h, np.nan), x[c].astype(np.float64), np.full(h, np.nan)])
            win = np.lib.stride_tricks.sliding_window_view(pad, W)
            with np.errstate(invalid="ignore"):
                cnt = (~np.isnan(win)).sum(1)
                m = np.where(cnt > 0, np.nansum(win, 1) / np.maximum(cnt, 1), np.nan)
            m = m.astype(x.dtype).astype(np.float64)  # the mean comes back in the data dtype
        else:
            m = x[c].astype(np.float64)
        for g in range(G):
            idx = np.nonzero(gidx == g)[0]
            if idx.size == 0:
                continue
            xs_, ys_ = tn[idx], m[idx]
            ok = ~np.isnan(ys_)
            if ok.sum() == 0:
                continue
            lhs = np.vander(xs_, degree + 1)
            scale = np.sqrt((lhs * lhs).sum(axis=0))
            lhs = lhs / scale
            coef, *_ = np.linalg.lstsq(lhs[ok], ys_[ok], rcond=xs_.shape[0] * np.finfo(np.float64).eps)
            coef = coef / scale
            tr = np.zeros(idx.size)
            for k in range(degree + 1):  # Horner, highest power first (np.vander order)
                tr = tr * xs_ + coef[k]
            if preserve_mean:
                tr = tr - tr.mean() if kind == "+" else tr / tr.mean()
            out[c, idx] = tr
    return out.astype(x.dtype)


def interp_on_quantiles_1d(newx, oldx, oldy, method="linear", extrap="constant"):
    """utils._interp_on_quantiles_1D (utils.py:350-377) for ONE cell with scipy.interpolate.interp1d itself: NaN pairs of the
    table dropped, NaN abscissae stay NaN, outside the table the first / last valid factor ("constant") or NaN."""
    from scipy.interpolate import interp1d

    newx, oldx, oldy = np.asarray(newx), np.asarray(oldx), np.asarray(oldy)
    mask_new = np.isnan(newx)
    mask_old = np.isnan(oldy) | np.isnan(oldx)
    out = np.full(newx.shape, np.nan, oldy.dtype)
    if mask_new.all() or mask_old.all():
        return out
    if extrap == "constant":  # the first / last non-NaN FACTOR, whatever its abscissa (utils.py:362-366)
        fill = (oldy[~np.isnan(oldy)][0], oldy[~np.isnan(oldy)][-1])
    else:
        fill = np.nan
    out[~mask_new] = interp1d(oldx[~mask_old], oldy[~mask_old], kind=method, bounds_error=False, fill_value=fill)(newx[~mask_new])
    return out
