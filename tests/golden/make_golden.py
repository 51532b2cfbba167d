#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the reference's OWN kernel source.

Runs only in the build container (needs /root/reference, read-only).  Nothing here
ships or is imported by the product or by the GPU-side tests: the output is data
(``*.npz``: inputs + expected outputs) and this script is committed so the vectors
can be regenerated.

How the reference is executed:
  * ``/root/reference/src/xsdba/nbutils.py`` is exec'd as a module after registering
    stub ``numba`` (``njit``/``guvectorize`` = identity decorators) and ``xarray``
    modules, which yields ``_nan_quantile_1d``, ``_quantile``,
    ``_first_and_last_nonnull``, ``_extrapolate_on_quantiles`` as plain numpy code.
  * ``_interp_on_quantiles_1D``, ``_interp_on_quantiles_2d``, ``equally_spaced_nodes``
    are AST-extracted from ``utils.py`` and run with the real scipy
    (``interp1d``/``griddata``) installed here.
  * the xarray orchestration that cannot run here (rolling-construct windows, groupby,
    ``Grouper.apply``, polyfit/polyval, rank) is composed in numpy exactly as
    ``_adjustment.py`` / ``base.py`` / ``detrending.py`` / ``utils.py`` describe it;
    each helper cites the lines it follows.

numba typing note: ``_nan_quantile_1d`` is called with ``alpha=beta=np.float64(1)`` so the
virtual index is evaluated in float64 from the dtype-rounded ``q`` as numba's type-based
promotion does (nbutils.py:133).
"""
from __future__ import annotations

import ast
import importlib.util
import sys
import types
import warnings
from pathlib import Path

import numpy as np
from scipy.interpolate import griddata, interp1d
from scipy.stats import rankdata

REF = Path("/root/reference/src/xsdba")
OUT = Path(__file__).resolve().parent


# --------------------------------------------------------------------------------------
# loading the reference kernels
# --------------------------------------------------------------------------------------
def _install_stubs():
    def _identity_decorator(*dargs, **dkwargs):
        if len(dargs) == 1 and callable(dargs[0]) and not dkwargs:
            return dargs[0]
        return lambda f: f

    class _T:
        def __getitem__(self, item):
            return self

        def __call__(self, *a, **k):
            return self

    nb = types.ModuleType("numba")
    nb.njit = _identity_decorator
    nb.guvectorize = _identity_decorator
    nb.float32 = nb.float64 = nb.boolean = _T()
    sys.modules["numba"] = nb
    xr = types.ModuleType("xarray")
    xr.DataArray = object
    xr.apply_ufunc = None
    xrc = types.ModuleType("xarray.core")
    xrcu = types.ModuleType("xarray.core.utils")
    xrc.utils = xrcu
    xr.core = xrc
    sys.modules["xarray"] = xr
    sys.modules["xarray.core"] = xrc
    sys.modules["xarray.core.utils"] = xrcu


def load_reference():
    _install_stubs()
    spec = importlib.util.spec_from_file_location("ref_nbutils", REF / "nbutils.py")
    nbu = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(nbu)

    wanted = {"_interp_on_quantiles_1D", "_interp_on_quantiles_2d", "equally_spaced_nodes"}
    tree = ast.parse((REF / "utils.py").read_text())
    ns = {
        "np": np,
        "interp1d": interp1d,
        "griddata": griddata,
        "warn": warnings.warn,
        "_extrapolate_on_quantiles": nbu._extrapolate_on_quantiles,
    }
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in wanted:
            node.decorator_list = []
            mod = ast.Module(body=[node], type_ignores=[])
            exec(compile(mod, str(REF / "utils.py"), "exec"), ns)  # noqa: S102
    return nbu, ns


NBU, UT = load_reference()


def ref_quantile_rows(arr2d: np.ndarray, q: np.ndarray) -> np.ndarray:
    """nbutils._wrapper_quantile1d (nbutils.py:198-203) with numba's float64 index typing."""
    qc = np.array(q, dtype=arr2d.dtype)  # nbutils.py:253
    out = np.empty((arr2d.shape[0], qc.size), dtype=arr2d.dtype)
    for i in range(arr2d.shape[0]):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            out[i] = NBU._nan_quantile_1d(arr2d[i].copy(), qc, np.float64(1), np.float64(1))
    return out


# --------------------------------------------------------------------------------------
# calendar / grouping helpers (noleap), restating base.py:232-345
# --------------------------------------------------------------------------------------
_DPM = np.array([31, 28, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31])


def noleap_fields(T: int, start_doy0: int = 0):
    """doy (1..365), month (1..12), day-of-month, year index for T consecutive noleap days."""
    d0 = (np.arange(T) + start_doy0) % 365
    year = (np.arange(T) + start_doy0) // 365
    cum = np.concatenate([[0], np.cumsum(_DPM)])
    month = np.searchsorted(cum, d0, side="right")
    day = d0 - cum[month - 1] + 1
    return d0 + 1, month, day, year


def time_ns_noleap(T: int, start_year: int = 2000, start_doy0: int = 0) -> np.ndarray:
    """Numeric time used by xarray.polyfit: ns since 1970-01-01 in the (noleap) calendar, float64."""
    days = (start_year - 1970) * 365 + start_doy0 + np.arange(T, dtype=np.int64)
    return (days * 86400 * 10**9).astype(np.float64)


def rolling_construct(x: np.ndarray, W: int) -> np.ndarray:
    """xarray rolling(center=True, time=W).construct('window') on the last axis (base.py:261-262).

    Row t, slot j holds x[t + j - W//2]; NaN outside [0, T).
    """
    T = x.shape[-1]
    out = np.full(x.shape + (W,), np.nan, dtype=x.dtype)
    for j in range(W):
        o = j - W // 2
        lo, hi = max(0, -o), min(T, T - o)
        out[..., lo:hi, j] = x[..., lo + o : hi + o]
    return out


def group_rows(x: np.ndarray, gidx: np.ndarray, G: int, W: int):
    """List (per group g=0..G-1) of the (cells, n_time_in_group*W) blocks Grouper.apply hands to func.

    base.py:261-272, 410-420: windows are constructed first, then grouped on the group of the
    CENTRE time; reduction dims are [time, window].
    """
    xr_ = rolling_construct(x, W) if W > 1 else x[..., None]
    rows = []
    for g in range(G):
        sel = np.nonzero(gidx == g)[0]
        rows.append(xr_[..., sel, :].reshape(x.shape[:-1] + (-1,)))
    return rows


def get_correction(x, y, kind):  # utils.py:130-143
    return y - x if kind == "+" else y / x


def apply_correction(x, f, kind):  # utils.py:146-162
    return x + f if kind == "+" else x * f


def invert(x, kind):  # utils.py:165-177
    return -x if kind == "+" else 1 / x


# --------------------------------------------------------------------------------------
# train bodies: _adjustment.py:193-286 (eqm_train), :95-190 (dqm_train)
# --------------------------------------------------------------------------------------
def eqm_train(ref, hist, gidx, G, W, q, kind):
    C = ref.shape[0]
    af = np.empty((C, G, q.size), ref.dtype)
    hq = np.empty_like(af)
    rr, hh = group_rows(ref, gidx, G, W), group_rows(hist, gidx, G, W)
    for g in range(G):
        ref_q = ref_quantile_rows(rr[g], q)
        hist_q = ref_quantile_rows(hh[g], q)
        with np.errstate(all="ignore"):
            af[:, g] = get_correction(hist_q, ref_q, kind)
        hq[:, g] = hist_q
    return af, hq


def dqm_train(ref, hist, gidx, G, W, q, kind):
    C = ref.shape[0]
    af = np.empty((C, G, q.size), ref.dtype)
    hq = np.empty_like(af)
    scaling = np.empty((C, G), ref.dtype)
    rr, hh = group_rows(ref, gidx, G, W), group_rows(hist, gidx, G, W)
    for g in range(G):
        with warnings.catch_warnings(), np.errstate(all="ignore"):
            warnings.simplefilter("ignore")
            mu_ref = np.nanmean(rr[g], axis=-1)  # xarray mean(skipna) on float32 -> np.nanmean
            mu_hist = np.nanmean(hh[g], axis=-1)
            refn = apply_correction(rr[g], invert(mu_ref, kind)[:, None], kind)  # :167
            histn = apply_correction(hh[g], invert(mu_hist, kind)[:, None], kind)  # :168
            ref_q = ref_quantile_rows(refn.astype(ref.dtype), q)
            hist_q = ref_quantile_rows(histn.astype(ref.dtype), q)
            af[:, g] = get_correction(hist_q, ref_q, kind)
            hq[:, g] = hist_q
            scaling[:, g] = get_correction(mu_hist, mu_ref, kind)  # :177-179
    return af, hq, scaling


# --------------------------------------------------------------------------------------
# adjust bodies
# --------------------------------------------------------------------------------------
def add_cyclic(tab):  # utils.py:284-314 (pad mode="wrap" along the group axis)
    return np.concatenate([tab[:, -1:], tab, tab[:, :1]], axis=1)


def interp_on_quantiles(newx, xq, yq, newg, G, method, extrap):
    """utils.interp_on_quantiles (utils.py:408-513) for cells on axis 0.

    newx (C,T); xq,yq (C,G,Q); newg (T,) group coordinate (1..G) or None when G==1 ('time').
    """
    C, T = newx.shape
    out = np.empty((C, T), dtype=yq.dtype)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        if newg is None:
            for c in range(C):
                out[c] = UT["_interp_on_quantiles_1D"](newx[c], xq[c, 0], yq[c, 0], method, extrap)
            return out
        xqe, yqe = add_cyclic(xq), add_cyclic(yq)
        Q = xq.shape[-1]
        oldg = np.repeat(np.arange(0, G + 2, dtype=np.int64)[:, None], Q, axis=1)  # coords 0..G+1
        for c in range(C):
            out[c] = UT["_interp_on_quantiles_2d"](newx[c], newg, xqe[c], yqe[c], oldg, method, extrap)
    return out


def qm_adjust(sim, af, hist_q, newg, G, interp, extrap, kind):  # _adjustment.py:594-676
    af_t = interp_on_quantiles(sim, hist_q, af, newg, G, interp, extrap)
    return apply_correction(sim, af_t, kind).astype(sim.dtype)


def window_mean_trend(x, gidx, G, W, tns, degree=1):
    """detrending.py:196-208 + xarray polyfit/polyval (float64 Vandermonde, column scaled, lstsq)."""
    C, T = x.shape
    trend = np.full((C, T), np.nan, dtype=np.float64)
    xr_ = rolling_construct(x, W) if W > 1 else x[..., None]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = np.nanmean(xr_, axis=-1) if W > 1 else x  # da.mean(dim[1:]) in the data dtype
    for g in range(G):
        sel = np.nonzero(gidx == g)[0]
        if sel.size == 0:
            continue
        tx = tns[sel]
        lhs = np.vander(tx, degree + 1)
        scale = np.sqrt((lhs * lhs).sum(axis=0))
        lhs = lhs / scale
        for c in range(C):
            rhs = m[c, sel].astype(np.float64)
            ok = ~np.isnan(rhs)
            if ok.sum() == 0:
                continue
            coef, *_ = np.linalg.lstsq(lhs[ok], rhs[ok], rcond=None)
            coef = coef / scale
            res = np.zeros_like(tx)
            for cf in coef:  # Horner, highest degree first (xr.polyval)
                res = res * tx + cf
            trend[c, sel] = res
    return trend.astype(x.dtype)  # detrending.py:86-87


def dqm_adjust(sim, af, hist_q, scaling, gidx, newg, G, W, tns, interp, extrap, kind, degree=1):
    """_adjustment.py:679-780 with a doy/'time' group (scaling broadcast is nearest, :748-757)."""
    sc_t = scaling[:, gidx]  # broadcast(sel(method="nearest")) on the integer group index
    scaled = apply_correction(sim, sc_t, kind).astype(sim.dtype)
    trend = window_mean_trend(scaled, gidx, G, W, tns, degree)
    with np.errstate(all="ignore"):
        detr = apply_correction(scaled, invert(trend, kind), kind).astype(sim.dtype)
        scen = qm_adjust(detr, af, hist_q, newg, G, interp, extrap, kind)
        scen = apply_correction(scen, trend, kind).astype(sim.dtype)
    return scen, trend


def pct_rank_rows(rows):
    """utils.rank(pct=True) (utils.py:573-638) over the last axis; bottleneck.nanrankdata == scipy
    rankdata(method='average', nan_policy='omit')."""
    with warnings.catch_warnings(), np.errstate(all="ignore"):
        warnings.simplefilter("ignore")
        r = rankdata(rows, axis=-1, nan_policy="omit").astype(np.float64)
        cnt = np.sum(~np.isnan(rows), axis=-1, keepdims=True)
        rnk = r / cnt  # xarray Variable.rank(pct=True)
        mn = np.nanmin(rnk, axis=-1, keepdims=True)
        mx = np.nanmax(rnk, axis=-1, keepdims=True)
        return mx * (rnk - mn) / (mx - mn)


def qdm_sim_q(sim, gidx, G, W, rank_window):
    """_adjustment.py:872: group.apply(u.rank, sim, main_only=not rank_window, pct=True)."""
    C, T = sim.shape
    out = np.full((C, T), np.nan, dtype=np.float64)
    for g in range(G):
        sel = np.nonzero(gidx == g)[0]
        if sel.size == 0:
            continue
        if rank_window and W > 1:
            blk = rolling_construct(sim, W)[:, sel, :]  # (C, n, W)
            pr = pct_rank_rows(blk.reshape(C, -1)).reshape(blk.shape)
            out[:, sel] = pr[:, :, W // 2]  # base.py:438-439
        else:
            out[:, sel] = pct_rank_rows(sim[:, sel])
    return out


def qdm_adjust(sim, af, quantiles, gidx, newg, G, W, interp, extrap, kind, rank_window):
    sim_q = qdm_sim_q(sim, gidx, G, W, rank_window)
    xq = np.broadcast_to(quantiles.astype(af.dtype), af.shape).copy()
    af_t = interp_on_quantiles(sim_q, xq, af, newg, G, interp, extrap)
    with np.errstate(all="ignore"):
        scen = apply_correction(sim, af_t, kind).astype(sim.dtype)
    return scen, sim_q


# --------------------------------------------------------------------------------------
# vector sets
# --------------------------------------------------------------------------------------
def gen_quantiles(rng):
    d = {}
    q20 = UT["equally_spaced_nodes"](20)
    q50 = np.linspace(0.1, 0.99, 50)
    i = 0
    for dt in (np.float32, np.float64):
        for L in (1, 2, 3, 30, 31, 64, 65, 257, 912, 930, 1024, 1860, 3650):
            for nanfrac in (0.0, 0.1, "all_but_one", "all"):
                for qq in (q20, q50):
                    a = (rng.standard_normal(L) * 8 + 280).astype(dt)
                    if nanfrac == "all":
                        a[:] = np.nan
                    elif nanfrac == "all_but_one":
                        keep = rng.integers(L)
                        v = a[keep]
                        a[:] = np.nan
                        a[keep] = v
                    elif nanfrac:
                        a[rng.random(L) < nanfrac] = np.nan
                    if L >= 30 and i % 5 == 0:  # ties
                        a[: L // 3] = np.round(a[: L // 3])
                    d[f"arr_{i}"] = a
                    d[f"q_{i}"] = qq
                    d[f"out_{i}"] = ref_quantile_rows(a[None, :], qq)[0]
                    i += 1
    d["n"] = np.array(i)
    # tests/test_nbutils.py:23-34 edge cases
    q = np.linspace(0.1, 0.99, 50)
    one = np.array([1.0] + [np.nan] * 100)
    d["edge_one_out"] = ref_quantile_rows(one[None], q)[0]
    d["edge_nan_out"] = ref_quantile_rows(np.full((1, 100), np.nan), q)[0]
    np.savez_compressed(OUT / "quantile_1d.npz", **d)
    return i


def gen_interp_1d(rng):
    d = {}
    i = 0
    # tests/test_utils.py:68-113 and :149-194 (known answers 4.4 / 2.9 / 2.95 / NaN)
    for with_nan_tail in (False, True):
        for punch in (False, True):
            xq = np.linspace(205, 229, num=25)
            yq = np.linspace(2, 4.4, num=25)
            if with_nan_tail:
                xq = np.append(xq, [np.nan] * 5)
                yq = np.append(yq, [np.nan] * 5)
            if punch:
                xq = np.where(xq != 220, xq, np.nan)
                yq = np.where(yq != 3, yq, np.nan)
            newx = np.linspace(240, 200, num=41) - 0.5
            newx = np.where(newx > 201, newx, np.nan)
            for method in ("nearest", "linear"):
                for extrap in ("constant", "nan"):
                    for dt in (np.float64, np.float32):
                        a, b, c = newx.astype(dt), xq.astype(dt), yq.astype(dt)
                        with warnings.catch_warnings():
                            warnings.simplefilter("ignore")
                            o = UT["_interp_on_quantiles_1D"](a, b, c, method, extrap)
                        d[f"newx_{i}"], d[f"xq_{i}"], d[f"yq_{i}"] = a, b, c
                        d[f"method_{i}"], d[f"extrap_{i}"] = np.array(method), np.array(extrap)
                        d[f"out_{i}"] = o
                        i += 1
    # random monotone tables, float32, some NaN holes, queries spanning beyond the table
    for _ in range(12):
        Q = int(rng.choice([1, 2, 5, 20, 50]))
        xq = np.sort(rng.standard_normal(Q) * 5 + 280).astype(np.float32)
        yq = (rng.standard_normal(Q)).astype(np.float32)
        if Q > 4 and rng.random() < 0.5:
            xq[rng.integers(Q)] = np.nan
            yq[rng.integers(Q)] = np.nan
        newx = (rng.standard_normal(300) * 7 + 280).astype(np.float32)
        newx[rng.random(300) < 0.05] = np.nan
        newx[:Q] = np.where(np.isnan(xq), newx[:Q], xq)  # exact hits on the nodes
        for method in ("nearest", "linear"):
            for extrap in ("constant", "nan"):
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    try:
                        o = UT["_interp_on_quantiles_1D"](newx, xq, yq, method, extrap)
                    except ValueError:
                        continue  # interp1d refuses (e.g. no valid node left for 'linear')
                d[f"newx_{i}"], d[f"xq_{i}"], d[f"yq_{i}"] = newx, xq, yq
                d[f"method_{i}"], d[f"extrap_{i}"] = np.array(method), np.array(extrap)
                d[f"out_{i}"] = o
                i += 1
    d["n"] = np.array(i)
    np.savez_compressed(OUT / "interp_1d.npz", **d)
    return i


def synth_temperature(rng, C, T, start_doy0=0, dtype=np.float32):
    """tests/test_utils.py:118-133 style seasonal series (SURVEY §8d cfg2/3 generator)."""
    doy, _, _, _ = noleap_fields(T, start_doy0)
    t = np.arange(T)
    off = rng.standard_normal((C, 1)) * 3
    ref = 280 - 20 * np.cos(2 * np.pi * doy / 365) + 2 * rng.random((C, T)) + 0.1 * t / 365 + off
    hist = 280 - 18 * np.cos(2 * np.pi * doy / 365) + 2 * rng.random((C, T)) + 0.11 * t / 365 + off
    sim = 281 - 18 * np.cos(2 * np.pi * doy / 365) + 2.2 * rng.random((C, T)) + 0.15 * t / 365 + off
    return ref.astype(dtype), hist.astype(dtype), sim.astype(dtype)


def synth_precip(rng, C, T, dtype=np.float32):
    """testing/helpers.py:23-35 (Cannon 2015) gamma distributions, strictly positive."""
    ref = rng.gamma(4, 7.5, (C, T))
    hist = rng.gamma(8.15, 3.68, (C, T))
    sim = rng.gamma(16, 2.63, (C, T))
    return ref.astype(dtype), hist.astype(dtype), sim.astype(dtype)


def gen_interp_2d(rng):
    """_interp_on_quantiles_2d (utils.py:380-400) on trained tables; month (G=12) and doy (G=365)."""
    d = {}
    i = 0
    T = 3650
    doy, month, _, _ = noleap_fields(T)
    q = UT["equally_spaced_nodes"](20).astype(np.float32)
    for name, gidx, G, W, data in (
        ("month", month - 1, 12, 1, synth_temperature(rng, 2, T)),
        ("doy", doy - 1, 365, 31, synth_temperature(rng, 2, T)),
        ("month_pr", month - 1, 12, 1, synth_precip(rng, 2, T)),
    ):
        ref, hist, sim = data
        kind = "*" if name.endswith("pr") else "+"
        af, hq = eqm_train(ref, hist, gidx, G, W, q, kind)
        sim = sim.copy()
        sim[0, rng.random(T) < 0.02] = np.nan
        if name == "month":  # a table with NaN holes and an all-NaN group row
            hq = hq.copy()
            af = af.copy()
            hq[1, 3, 5] = np.nan
            af[1, 7, 0] = np.nan
            hq[1, 9, :] = np.nan
        newg = (gidx + 1).astype(np.int64)
        for extrap in ("constant", "nan"):
            o = interp_on_quantiles(sim, hq, af, newg, G, "nearest", extrap)
            d[f"sim_{i}"], d[f"hq_{i}"], d[f"af_{i}"], d[f"newg_{i}"] = sim, hq, af, newg
            d[f"G_{i}"], d[f"extrap_{i}"], d[f"out_{i}"] = np.array(G), np.array(extrap), o
            i += 1
    d["n"] = np.array(i)
    np.savez_compressed(OUT / "interp_2d_nearest.npz", **d)
    return i


def gen_rank(rng):
    d = {}
    i = 0
    for L in (1, 2, 7, 30, 31, 930):
        for ties in (False, True):
            for nanfrac in (0.0, 0.2):
                a = rng.standard_normal((3, L)).astype(np.float32)
                if ties:
                    a = np.round(a * 2) / 2
                if nanfrac:
                    a[rng.random(a.shape) < nanfrac] = np.nan
                d[f"arr_{i}"] = a
                d[f"out_{i}"] = pct_rank_rows(a)
                i += 1
    d["n"] = np.array(i)
    # tests/test_utils.py:208-217 tie example (integer ranks, average ties)
    arr = np.array([1, 26, 2, 4.0, 6, 2, 2])
    d["tie_arr"] = arr
    d["tie_ranks"] = rankdata(arr)
    np.savez_compressed(OUT / "rank.npz", **d)
    return i


def gen_pipelines(rng):
    """Full train+adjust on a few cells x 3650 noleap days, composed as _adjustment.py does."""
    n = 0
    T, C = 3650, 3
    doy, month, _, _ = noleap_fields(T)
    tns = time_ns_noleap(T)
    q20 = UT["equally_spaced_nodes"](20).astype(np.float32)
    groupings = {
        "time": (np.zeros(T, np.int64), 1, 1),
        "month": (month - 1, 12, 1),
        "doy31": (doy - 1, 365, 31),
        "doy1": (doy - 1, 365, 1),
        "month5": (month - 1, 12, 5),  # window>1 with a coarse group -> multiplicities (base.py:261-272)
    }
    for gname, (gidx, G, W) in groupings.items():
        for kind in ("+", "*"):
            ref, hist, sim = synth_temperature(rng, C, T) if kind == "+" else synth_precip(rng, C, T)
            if gname == "doy31":  # NaN handling: one all-NaN cell, sprinkled NaNs elsewhere
                ref[2, :] = np.nan
                hist[2, :] = np.nan
                sim[2, :] = np.nan
                for a in (ref, hist, sim):
                    a[1, rng.random(T) < 0.01] = np.nan
            newg = None if G == 1 else (gidx + 1).astype(np.int64)
            d = dict(ref=ref, hist=hist, sim=sim, gidx=gidx.astype(np.int32), G=np.array(G), W=np.array(W),
                     q=q20, kind=np.array(kind), tns=tns)
            # --- EQM / QDM share the train (adjustment.py:674)
            af, hq = eqm_train(ref, hist, gidx, G, W, q20, kind)
            d["eqm_af"], d["eqm_hist_q"] = af, hq
            for interp in ("nearest",) + (("linear",) if G == 1 else ()):
                for extrap in ("constant", "nan"):
                    with np.errstate(all="ignore"):
                        d[f"eqm_scen_{interp}_{extrap}"] = qm_adjust(sim, af, hq, newg, G, interp, extrap, kind)
            if gname in ("time", "month", "doy31", "doy1"):
                # --- QDM
                for rw in (False, True):
                    for interp in ("nearest",) + (("linear",) if G == 1 else ()):
                        scen, sim_q = qdm_adjust(sim, af, q20, gidx, newg, G, W, interp, "constant", kind, rw)
                        d[f"qdm_scen_{interp}_rw{int(rw)}"] = scen
                        d[f"qdm_simq_rw{int(rw)}"] = sim_q
            if gname in ("time", "doy31", "doy1", "month"):
                # --- DQM
                daf, dhq, dsc = dqm_train(ref, hist, gidx, G, W, q20, kind)
                d["dqm_af"], d["dqm_hist_q"], d["dqm_scaling"] = daf, dhq, dsc
                for interp in ("nearest",) + (("linear",) if G == 1 else ()):
                    scen, trend = dqm_adjust(sim, daf, dhq, dsc, gidx, newg, G, W, tns, interp, "constant", kind)
                    d[f"dqm_scen_{interp}"] = scen
                    d["dqm_trend"] = trend
            np.savez_compressed(OUT / f"pipeline_{gname}_{'add' if kind == '+' else 'mul'}.npz", **d)
            n += 1
    return n


def gen_known_answers(rng):
    """Inputs of the reference's analytic tests so the oracle/HIP path can be held to the same
    assertions (tests/test_adjustment.py:947-989 TestQM.test_quantiles, :700-744 TestQDM)."""
    from scipy.stats import norm, uniform

    u = rng.random(10000)
    x = uniform(loc=10, scale=1).ppf(u)
    y = norm(loc=12, scale=1).ppf(u)
    q50 = UT["equally_spaced_nodes"](50)
    d = dict(u=u, x=x, y=y, q50=q50)
    gidx = np.zeros(10000, np.int64)
    for kind, tag in (("+", "add"), ("*", "mul")):
        af, hq = eqm_train(y[None], x[None], gidx, 1, 1, q50, kind)
        d[f"eqm_af_{tag}"], d[f"eqm_hq_{tag}"] = af, hq
        d[f"eqm_p_linear_{tag}"] = qm_adjust(x[None], af, hq, None, 1, "linear", "constant", kind)
    np.savez_compressed(OUT / "known_answer_qm.npz", **d)
    return 1


def main():
    rng = np.random.default_rng(20260101)
    print("quantile_1d cases:", gen_quantiles(rng))
    print("interp_1d cases:", gen_interp_1d(rng))
    print("interp_2d cases:", gen_interp_2d(rng))
    print("rank cases:", gen_rank(rng))
    print("pipelines:", gen_pipelines(rng))
    print("known answers:", gen_known_answers(rng))


if __name__ == "__main__":
    main()
