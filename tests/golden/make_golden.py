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

    wanted = {"_interp_on_quantiles_1D", "_interp_on_quantiles_2d", "equally_spaced_nodes", "_ecdf_1d", "map_cdf_1d"}
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


# --------------------------------------------------------------------------------------
# round 2: the rows of the hot path's neighbourhood that had no reference-generated vector
# --------------------------------------------------------------------------------------
def month_fractional_index(T, start_doy0=0):
    """Grouper.get_index(interp=True) for time.month on a noleap axis (base.py:308-309):
    month - 0.5 + day / days_in_month."""
    _, month, day, _ = noleap_fields(T, start_doy0)
    return month - 0.5 + day / _DPM[month - 1]


def broadcast_linear(tab, i):
    """utils.broadcast(grouped, x, interp="linear") for a month table (utils.py:222-246): cyclic bounds without cyclic
    coordinates (months 0 and 13, :284-314), then DataArray.interp(method="linear") = numpy.interp per cell in float64,
    cast back to the table dtype (:243)."""
    C, G = tab.shape
    ext = np.concatenate([tab[:, -1:], tab, tab[:, :1]], axis=1)
    xp = np.arange(0, G + 2, dtype=np.float64)
    out = np.empty((C, i.size), tab.dtype)
    for c in range(C):
        out[c] = np.interp(i, xp, ext[c].astype(np.float64)).astype(tab.dtype)
    return out


def group_nanmean(x, gidx, G, W):
    """Grouper.apply("mean", x, skipna=True) (base.py:347-457 with a string func): mean over (time, window) of every
    group block, in the data dtype (xarray -> numpy nanmean)."""
    rows = group_rows(x, gidx, G, W)
    out = np.empty((x.shape[0], G), x.dtype)
    with warnings.catch_warnings(), np.errstate(all="ignore"):
        warnings.simplefilter("ignore")
        for g in range(G):
            out[:, g] = np.nanmean(rows[g], axis=-1)
    return out


def scaling_train(ref, hist, gidx, G, W, kind):  # _adjustment.py:941-958
    return get_correction(group_nanmean(hist, gidx, G, W), group_nanmean(ref, gidx, G, W), kind)


def loci_train(ref, hist, gidx, G, W, thresh):  # _adjustment.py:889-914
    C = ref.shape[0]
    rr, hh = group_rows(ref, gidx, G, W), group_rows(hist, gidx, G, W)
    s_thresh = np.empty((C, G), hist.dtype)
    with warnings.catch_warnings(), np.errstate(all="ignore"):
        warnings.simplefilter("ignore")
        for g in range(G):
            for c in range(C):
                # group.apply(u.map_cdf, {x: hist, y: ref}, y_value=thresh).isel(x=0); output_dtypes=[x.dtype] (utils.py:74-84)
                s_thresh[c, g] = UT["map_cdf_1d"](hh[g][c], rr[g][c], np.atleast_1d(thresh))[0]
        sth = s_thresh[:, gidx]                                   # u.broadcast(s_thresh, hist, group) — nearest
        ws = np.where(hist >= sth, hist, np.nan).astype(hist.dtype)
        wo = np.where(ref >= thresh, ref, np.nan).astype(ref.dtype)
        ms, mo = group_nanmean(ws, gidx, G, W), group_nanmean(wo, gidx, G, W)
        af = get_correction(ms - s_thresh, mo - thresh, "*")
    return af.astype(ref.dtype), s_thresh


def loci_adjust(sim, af_t, sth_t, thresh):  # _adjustment.py:917-938 with the tables already broadcast on time
    with np.errstate(all="ignore"):
        return np.clip(af_t * (sim - sth_t) + thresh, 0, None).astype(sim.dtype)


def gen_families(rng):
    """Scaling, LOCI and the max_tail_factor mask (_adjustment.py:889-974, 648-658, 672-673)."""
    d = {}
    T, C = 3650, 3
    doy, month, _, _ = noleap_fields(T)
    imon = month_fractional_index(T)
    d["month_index_interp"] = imon
    q20 = UT["equally_spaced_nodes"](20).astype(np.float32)
    for gname, gidx, G, W in (("month", month - 1, 12, 1), ("doy31", doy - 1, 365, 31), ("time", np.zeros(T, np.int64), 1, 1)):
        d[f"{gname}_gidx"] = gidx.astype(np.int32)
        d[f"{gname}_GW"] = np.array([G, W])
        for kind, tag in (("+", "add"), ("*", "mul")):
            ref, hist, sim = synth_temperature(rng, C, T) if kind == "+" else synth_precip(rng, C, T)
            for a in (ref, hist, sim):
                a[1, rng.random(T) < 0.01] = np.nan
            k = f"{gname}_{tag}"
            d[f"{k}_ref"], d[f"{k}_hist"], d[f"{k}_sim"] = ref, hist, sim
            # --- Scaling
            af = scaling_train(ref, hist, gidx, G, W, kind)
            d[f"{k}_scaling_af"] = af
            with np.errstate(all="ignore"):
                d[f"{k}_scaling_scen_nearest"] = apply_correction(sim, af[:, gidx], kind).astype(sim.dtype)
                if gname == "month":
                    d[f"{k}_scaling_scen_linear"] = apply_correction(sim, broadcast_linear(af, imon), kind).astype(sim.dtype)
            # --- max_tail_factor mask on an EQM adjustment (hist_q_raw = hist_q without adapt_freq, _adjustment.py:262-270)
            if gname != "time":
                eaf, ehq = eqm_train(ref, hist, gidx, G, W, q20, kind)
                newg = (gidx + 1).astype(np.int64)
                with np.errstate(all="ignore"):
                    scen = qm_adjust(sim, eaf, ehq, newg, G, "nearest", "constant", kind)
                d[f"{k}_eqm_hist_q"], d[f"{k}_eqm_scen"] = ehq, scen
                for f in (1.0, 1.02):
                    last = ehq[:, :, -1][:, gidx]  # broadcast(last_quantile, sim, group, "nearest")
                    with np.errstate(all="ignore"):
                        mask = sim > np.float32(f) * last if sim.dtype == np.float32 else sim > f * last
                    d[f"{k}_tailmask_{f}"] = np.where(mask, sim, scen).astype(sim.dtype)
            # --- LOCI (multiplicative data only: a wet-day threshold)
            if kind == "*":
                thresh = 12.0
                laf, lth = loci_train(ref, hist, gidx, G, W, thresh)
                d[f"{k}_loci_af"], d[f"{k}_loci_hist_thresh"], d[f"{k}_loci_thresh"] = laf, lth, np.array(thresh)
                d[f"{k}_loci_scen_nearest"] = loci_adjust(sim, laf[:, gidx], lth[:, gidx], thresh)
                if gname == "month":
                    d[f"{k}_loci_scen_linear"] = loci_adjust(sim, broadcast_linear(laf, imon), broadcast_linear(lth, imon), thresh)
    np.savez_compressed(OUT / "families_scaling_loci_tail.npz", **d)
    return len(d)


def gen_vecquantiles(rng):
    """nbutils._vecquantiles (nbutils.py:151-161), run through the stub guvectorize: res[0] = np.nanquantile(arr, rnk)."""
    d = {}
    i = 0
    for dt in (np.float32, np.float64):
        for L in (1, 2, 3, 30, 930):
            a = (rng.standard_normal((12, L)) * 8 + 280).astype(dt)
            a[rng.random(a.shape) < 0.1] = np.nan
            a[3] = np.nan
            rnk = rng.random(12).astype(dt)
            rnk[0], rnk[1], rnk[2] = 0.0, 1.0, np.nan
            out = np.empty(12, dt)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                for r in range(12):
                    res = np.empty(1, dt)
                    NBU._vecquantiles(a[r], rnk[r], res)
                    out[r] = res[0]
            d[f"arr_{i}"], d[f"rnk_{i}"], d[f"out_{i}"] = a, rnk, out
            i += 1
    d["n"] = np.array(i)
    np.savez_compressed(OUT / "vecquantiles.npz", **d)
    return i


def gen_interp_2d_fractional(rng):
    """_interp_on_quantiles_2d (utils.py:380-400) with the FRACTIONAL group coordinate of Grouper.get_index(interp=True)
    (base.py:308-309) on monthly tables — what interp="linear" / "cubic" hand to griddata (utils.py:476-513)."""
    d = {}
    i = 0
    T = 730
    _, month, _, _ = noleap_fields(T)
    newg = month_fractional_index(T)
    q = UT["equally_spaced_nodes"](20).astype(np.float32)
    for name, data in (("tas", synth_temperature(rng, 2, 3650)), ("pr", synth_precip(rng, 2, 3650))):
        ref, hist, sim = data
        kind = "*" if name == "pr" else "+"
        _, month_l, _, _ = noleap_fields(3650)
        af, hq = eqm_train(ref, hist, month_l - 1, 12, 1, q, kind)
        if name == "tas":  # holes in the table
            hq, af = hq.copy(), af.copy()
            hq[1, 3, 5] = np.nan
            af[1, 7, 0] = np.nan
        sim = sim[:, :T].copy()
        sim[0, rng.random(T) < 0.02] = np.nan
        for method in ("nearest", "linear", "cubic"):
            for extrap in ("constant", "nan"):
                o = interp_on_quantiles(sim, hq, af, newg, 12, method, extrap)
                d[f"sim_{i}"], d[f"hq_{i}"], d[f"af_{i}"], d[f"newg_{i}"] = sim, hq, af, newg
                d[f"method_{i}"], d[f"extrap_{i}"], d[f"out_{i}"] = np.array(method), np.array(extrap), o
                i += 1
    d["n"] = np.array(i)
    np.savez_compressed(OUT / "interp_2d_fractional.npz", **d)
    return i


def gen_window_handcase():
    """The reference's own Grouper.apply expectation (tests/test_base.py:76-120): x = arange(366) on 2000-01-01.. (a leap
    year, standard calendar), group time.month, plain and with window=5: rolling(center=True).construct + groupby mean over
    (time, window).  Written with explicit loops over real dates — independent of rolling_construct / group_rows above and
    of the oracle's row gathering — so that the window / group bookkeeping itself has a hand-checkable vector:
    January without a window is mean(0..30) = 15."""
    import datetime

    start = datetime.date(2000, 1, 1)
    dates = [start + datetime.timedelta(days=k) for k in range(366)]
    x = np.arange(366, dtype=np.float64)
    month0 = np.array([dt.month - 1 for dt in dates], np.int32)
    plain = np.zeros(12)
    win5 = np.zeros(12)
    for m in range(12):
        ts = [k for k in range(366) if month0[k] == m]
        plain[m] = sum(x[k] for k in ts) / len(ts)
        vals = [x[k + o] for k in ts for o in (-2, -1, 0, 1, 2) if 0 <= k + o < 366]
        win5[m] = sum(vals) / len(vals)
    assert plain[0] == 15.0 and plain[1] == 45.0  # January: days 0..30, February 2000: days 31..59
    d = dict(x=x, gidx=month0, plain=plain, win5=win5)
    # the same series, day-of-year groups with window 5 (test_base.py:31-44: every group holds one centre x 5 slots)
    doy0 = np.arange(366, dtype=np.int32)
    d["doy_gidx"] = doy0
    d["doy_win5"] = np.array([np.mean([x[k + o] for o in (-2, -1, 0, 1, 2) if 0 <= k + o < 366]) for k in range(366)])
    np.savez_compressed(OUT / "window_groups_handcase.npz", **d)
    return 1


def ecdf_rows(rows, value):
    """utils.ecdf (utils.py:87-105) on a (cells, n) block: (x <= value).sum(dim) / x.notnull().sum(dim)."""
    with np.errstate(invalid="ignore", divide="ignore"):
        return (rows <= value).sum(axis=-1) / (~np.isnan(rows)).sum(axis=-1)


def gen_preproc_deterministic(rng):
    """The deterministic half of the two pre-processing steps whose replacement draws are random in the reference:

    _processing._adapt_freq (_processing.py:75-142): P0_sim / P0_hist = ecdf(sim, thresh) (:90-92, utils.py:105), P0_ref,
    dP0 = where(P0_hist == 0, nan, (P0_hist - P0_ref) / P0_hist) (:94), pth = nbutils.vecquantiles(ref, P0_hist).where(dP0 > 0)
    (:103-105: the reference's own _vecquantiles kernel, executed here), and — for groups without a window — the NUMBER of
    replaced values per (cell, group): the percent ranks of utils.rank(pct=True, use_random_tiebreak=True) are the set
    {(k - 1) / (N - 1)} whatever the tie-break draws (utils.py:622-635), so the count of ranks inside
    [(P0_ref / P0_hist) * P0_sim, P0_sim] (:112) is fixed; only WHICH of the tied values are replaced, and by what, is random.

    processing.jitter (processing.py:178-257): the bounds after np.nextafter (:217-219, 239-243, the very numpy calls) and the
    masks of the values that are replaced (:228, 252)."""
    d = {}
    T = 3650
    _, month, _, _ = noleap_fields(T)
    cases = []
    for dt in (np.float32, np.float64):
        ref = (rng.gamma(0.8, 6.0, (3, T))).astype(dt)
        sim = (rng.gamma(0.9, 5.0, (3, T))).astype(dt)
        ref[rng.random((3, T)) < 0.25] = 0.0
        sim[rng.random((3, T)) < 0.45] = 0.0            # more dry days in the model than observed: dP0 > 0
        sim[1, month == 3] = np.abs(sim[1, month == 3]) + 2.0   # a group without dry days in sim: P0_hist = 0 -> dP0 NaN
        ref[2, month == 7] = 0.0                                   # more dry days in ref: dP0 < 0, nothing is adapted
        sim[0, rng.random(T) < 0.02] = np.nan
        ref[0, rng.random(T) < 0.02] = np.nan
        cases.append((ref, sim))
    i = 0
    for ref, sim in cases:
        for gname, gidx, G, W in (("month", month - 1, 12, 1), ("time", np.zeros(T, np.int32), 1, 1), ("month_w5", month - 1, 12, 5)):
            thresh = 1.0
            rs, rr = group_rows(sim, gidx, G, W), group_rows(ref, gidx, G, W)
            P0_hist = np.stack([ecdf_rows(r, thresh) for r in rs], axis=1)      # [cell, group] (P0_sim: the training branch)
            P0_ref = np.stack([ecdf_rows(r, thresh) for r in rr], axis=1)
            with np.errstate(invalid="ignore", divide="ignore"):
                dP0 = np.where(P0_hist == 0, np.nan, (P0_hist - P0_ref) / P0_hist)
            pth = np.full(P0_hist.shape, np.nan, sim.dtype)
            nrep = np.zeros(P0_hist.shape, np.int64)
            for c in range(sim.shape[0]):
                for g in range(G):
                    if dP0[c, g] > 0:
                        res = np.empty(1, sim.dtype)
                        with warnings.catch_warnings():
                            warnings.simplefilter("ignore")
                            NBU._vecquantiles(rr[g][c], np.asarray(P0_hist[c, g], sim.dtype), res)   # nbutils.py:151-161
                        pth[c, g] = res[0]
                        row = rs[g][c]
                        N = int((~np.isnan(row)).sum())
                        if N > 1:
                            rk = np.arange(N) / (N - 1.0)   # the percent ranks of the N valid values, whatever the tie-break
                            nrep[c, g] = int(((rk >= (P0_ref[c, g] / P0_hist[c, g]) * P0_hist[c, g]) & (rk <= P0_hist[c, g])).sum())
            k = f"af{i}"
            d[f"{k}_ref"], d[f"{k}_sim"], d[f"{k}_gidx"], d[f"{k}_GW"], d[f"{k}_thresh"] = ref, sim, gidx.astype(np.int32), np.array([G, W]), np.array(thresh)
            d[f"{k}_P0_ref"], d[f"{k}_P0_hist"], d[f"{k}_dP0"], d[f"{k}_pth"], d[f"{k}_nrep"] = P0_ref, P0_hist, dP0, pth, nrep
            i += 1
    d["n_af"] = np.array(i)
    # ---- jitter
    j = 0
    for dt in (np.float32, np.float64):
        x = (rng.gamma(0.8, 2.0, (2, 500))).astype(dt)
        x[0, rng.random(500) < 0.05] = np.nan
        x[1, :5] = [0.0, 1.0, 2.0, 3.0, 0.999]
        for lower, minimum, upper, maximum in ((1.0, None, None, None), (None, None, 2.0, 3.0), (1.0, 0.25, 2.0, 3.0)):
            k = f"jit{j}"
            notnull = ~np.isnan(x)
            d[f"{k}_x"] = x
            d[f"{k}_args"] = np.array([np.nan if v is None else v for v in (lower, minimum, upper, maximum)], np.float64)
            if lower is not None:
                jitter_lower = np.array(lower).astype(float)                                   # processing.py:216
                jitter_min = np.array(minimum if minimum is not None else 0).astype(float)     # :217
                jitter_min = np.nextafter(jitter_min.astype(x.dtype), np.inf, dtype=x.dtype)   # :218
                with np.errstate(invalid="ignore"):
                    d[f"{k}_mask_lower"] = (x < jitter_lower) & notnull                        # :228
                d[f"{k}_jitter_min"] = np.array(jitter_min, np.float64)
            if upper is not None:
                jitter_upper = np.array(upper).astype(float)                                   # :236
                jitter_max = np.array(maximum).astype(float)                                   # :237
                if x.dtype.itemsize < 8:                                                       # :240-241
                    jitter_max = np.nextafter(jitter_max.astype(x.dtype), -np.inf, dtype=x.dtype)
                with np.errstate(invalid="ignore"):
                    d[f"{k}_mask_upper"] = (x >= jitter_upper) & notnull                       # :252
                d[f"{k}_jitter_max"] = np.array(jitter_max, np.float64)
            j += 1
    d["n_jit"] = np.array(j)
    np.savez_compressed(OUT / "preproc_deterministic.npz", **d)
    return i + j


def gen_window_handcase_nonreducing():
    """The NON-reducing path of Grouper.apply (base.py:347-457) on the reference's own test series layout
    (tests/test_base.py:122-166: 366 days from 2000-01-01, group time.month, window 5): a function that keeps the grouped
    dimensions gets the windowed block of its group, the result keeps the window's CENTRE slot (base.py:438-439) and is put back
    in time order (:442-443).  Two such functions sit on the hot path — utils.rank (utils.py:573-638: average-tie ranks over
    [time, window], / N, rescaled mx * (r - mn) / (mx - mn)) and the per-group trend of PolyDetrend (detrending.py:196-208:
    mean over the window, skipna, then a least-squares line in time) — and both are written here with explicit loops over
    real dates, independent of rolling_construct / group_rows above and of the oracle's row gathering."""
    import datetime

    start = datetime.date(2000, 1, 1)
    dates = [start + datetime.timedelta(days=k) for k in range(366)]
    x = np.array([(k * 37) % 101 + 0.25 * (k % 3) for k in range(366)], np.float64)
    x[100] = x[103]  # ties, also across window slots
    x[200] = np.nan
    month0 = np.array([dt.month - 1 for dt in dates], np.int32)
    W = 5
    rank_pct = np.full(366, np.nan)
    trend = np.full(366, np.nan)
    day = np.array([(dt - datetime.date(1970, 1, 1)).days for dt in dates], np.float64)
    for m in range(12):
        ts = [k for k in range(366) if month0[k] == m]
        # ---- rank over [time, window]: every (centre k, slot o) pair of the group, out-of-range slots are NaN (skipped)
        V = np.array([x[k + o] for k in ts for o in range(-(W // 2), W // 2 + 1) if 0 <= k + o < 366])
        V = V[~np.isnan(V)]
        N = V.size

        def pct(v):
            return ((V < v).sum() + ((V == v).sum() + 1) / 2.0) / N

        allp = np.array([pct(v) for v in V])
        mn, mx = allp.min(), allp.max()
        for k in ts:
            if not np.isnan(x[k]):
                rank_pct[k] = mx * (pct(x[k]) - mn) / (mx - mn)
        # ---- trend: mean over the window at every centre (skipna), least-squares line against time
        mv = np.array([np.nanmean([x[k + o] for o in range(-(W // 2), W // 2 + 1) if 0 <= k + o < 366]) for k in ts])
        tt = day[ts]
        tc = tt - tt.mean()
        slope = (tc * (mv - mv.mean())).sum() / (tc * tc).sum()
        for k, t_ in zip(ts, tc):
            trend[k] = mv.mean() + slope * t_
    d = dict(x=x, gidx=month0, W=np.array(W), rank_pct=rank_pct, trend=trend, time_ns=day * 86400e9)
    np.savez_compressed(OUT / "window_groups_handcase_nonreducing.npz", **d)
    return 1


def gen_week_season_handcase():
    """Hand-checkable vectors for the two groupings that had none: `time.season` and `time.week` (base.py:196-198, 312-327).
    x = arange(731) on 2000-01-01 .. 2001-12-31 (standard calendar).  Everything comes from python's own `datetime` — the
    ISO week of `date.isocalendar()`, the season of the month — with explicit loops, independent of xsdba_amd.calendar, of
    the oracle's row gathering and of rolling_construct above: winter 2000 without a window is the mean over January,
    February and December 2000 plus January, February and December 2001."""
    import datetime

    start = datetime.date(2000, 1, 1)
    n = 731
    dates = [start + datetime.timedelta(days=k) for k in range(n)]
    assert dates[-1] == datetime.date(2001, 12, 31)
    x = np.arange(n, dtype=np.float64)
    season0 = np.array([dt.month % 12 // 3 for dt in dates], np.int32)          # DJF = 0, MAM = 1, JJA = 2, SON = 3
    week0 = np.array([dt.isocalendar()[1] - 1 for dt in dates], np.int32)        # ISO week 1..53 -> 0..52
    assert week0[0] == 51 and week0[2] == 0          # Saturday 2000-01-01 is in ISO week 52 of 1999; Monday 2000-01-03 opens week 1
    assert season0[0] == 0 and season0[59] == 0 and season0[60] == 1   # 2000-02-29 is winter, 2000-03-01 spring

    def means(gidx, G, W):
        out = np.full(G, np.nan)
        h = W // 2
        for g in range(G):
            vals = [x[k + o] for k in range(n) if gidx[k] == g for o in range(-h, W - h) if 0 <= k + o < n]
            if vals:
                out[g] = sum(vals) / len(vals)
        return out

    d = dict(x=x, season_gidx=season0, week_gidx=week0, season_plain=means(season0, 4, 1), season_win5=means(season0, 4, 5),
             week_plain=means(week0, 53, 1), week_win3=means(week0, 53, 3))
    djf = [k for k in range(n) if dates[k].month in (12, 1, 2)]
    assert d["season_plain"][0] == sum(djf) / len(djf) and len(djf) == 31 + 29 + 31 + 31 + 28 + 31
    # ISO week 53 does not occur in 2000 / 2001 (2004 and 2009 have one): its group stays empty -> NaN
    assert np.isnan(d["week_plain"][52]) and not np.isnan(d["week_plain"][51])
    # the median of a season's row of arange, by hand: the middle of its sorted members
    med = np.full(4, np.nan)
    for g in range(4):
        m = sorted(k for k in range(n) if season0[k] == g)
        med[g] = (m[(len(m) - 1) // 2] + m[len(m) // 2]) / 2.0
    d["season_median"] = med
    np.savez_compressed(OUT / "week_season_handcase.npz", **d)
    return 1


def main_round4():
    print("week / season handcase:", gen_week_season_handcase())


def main_round3():
    rng = np.random.default_rng(20260303)
    print("pre-processing, deterministic half:", gen_preproc_deterministic(rng))
    print("window handcase, non-reducing functions:", gen_window_handcase_nonreducing())


def main_round2():
    rng = np.random.default_rng(20260202)
    print("families:", gen_families(rng))
    print("vecquantiles:", gen_vecquantiles(rng))
    print("interp_2d fractional:", gen_interp_2d_fractional(rng))
    print("window handcase:", gen_window_handcase())


def main():
    rng = np.random.default_rng(20260101)
    print("quantile_1d cases:", gen_quantiles(rng))
    print("interp_1d cases:", gen_interp_1d(rng))
    print("interp_2d cases:", gen_interp_2d(rng))
    print("rank cases:", gen_rank(rng))
    print("pipelines:", gen_pipelines(rng))
    print("known answers:", gen_known_answers(rng))


if __name__ == "__main__":
    # `make_golden.py` regenerates the round-1 files, `make_golden.py round2` / `round3` / `round4` the files added in those rounds (own seeds, so that
    # neither run disturbs the other's vectors)
    if len(sys.argv) > 1 and sys.argv[1] == "round2":
        main_round2()
    elif len(sys.argv) > 1 and sys.argv[1] == "round3":
        main_round3()
    elif len(sys.argv) > 1 and sys.argv[1] == "round4":
        main_round4()
    else:
        main()
