"""The compute functions behind the adjustment classes, with the reference's names and signatures
(xsdba._adjustment: eqm_train :193-286, dqm_train :95-190, qm_adjust :594-676, dqm_adjust :679-780,
qdm_adjust :783-886).  Each is declared through ``map_groups`` / ``map_blocks`` like the reference, but its
body is ONE call into the HIP library that covers every group of every cell of the block.

Nothing here computes on the CPU: the few option combinations that are not on the device path raise NotImplementedError and
say so in their message (grouped ``interp="linear"`` / ``"cubic"`` ARE on it — utils._interp_on_quantiles_2d's griddata
interpolants, ``xsdba_hip_qm_adjust_linear2d`` / ``_cubic2d``, for EQM, DQM and QDM alike).

``Grouper(add_dims=...)`` (base.py:413: the extra dimensions are pooled into every group's sample at training, and into
the windowed ranks / the detrending fit at adjustment) needs no kernel of its own: the P pooled members of a cell are
laid end to end as ONE series of P * (T + window // 2) steps, the gaps filled with NaN and given no group, so that a
window never reaches from one member into the next; every training kernel then sees the pooled sample as the group's
row (NaN-skipping, like the out-of-range window slots of the reference's rolling.construct).
On the device since round 1: ``max_tail_factor`` (``hist_q_raw`` at training, ``hip.tail_mask`` after the adjustment),
``jitter_*`` and ``adapt_freq_thresh`` (``hip.jitter`` / ``hip.adapt_freq`` in front of the training kernels, the
frequency adaptation of ``sim`` with the trained ``P0_ref, P0_hist, pth`` in front of the adjust kernels).
"""
from __future__ import annotations

from dataclasses import dataclass

import warnings

import numpy as np

from . import hip
from . import nbutils as nbu
from . import utils as u
from .base import Grouper, map_blocks, map_groups
from .detrending import BaseDetrend, MeanDetrend, NoDetrend, PolyDetrend, RollingMeanDetrend
from .labeled import Coordinate, DataArray, Dataset, as_dataarray, full_like
from .utils import cell_time_pair, cell_time_view, table_view


def _unsupported(**opts):
    bad = {k: v for k, v in opts.items() if v is not None}
    if bad:
        raise NotImplementedError(
            f"option(s) {sorted(bad)} are not implemented on the HIP path yet (pre-processing rows of SURVEY §8f)"
        )


def _check_group(group: Grouper, ds):
    if group.add_dims:  # base.py:628-629
        if "sim" in ds and "af" in ds:  # adjust: sim need not carry the pooled dims
            return
        have = set()
        for name in ("ref", "hist", "sim"):
            if name in ds:
                have |= set(as_dataarray(ds[name]).dims)
        if not set(group.add_dims).issubset(have):
            raise ValueError("`add_dims` argument needs to be a dimension in one of the input datasets.")


class _Members:
    """[cell, member, time] view of an array whose ``group.add_dims`` dimensions are pooled (members) and whose other
    non-time dimensions are cells."""

    def __init__(self, da: DataArray, group: Grouper, dt):
        pooled = tuple(d for d in group.add_dims if d in da.dims)
        self.cell_dims = tuple(d for d in da.dims if d != group.dim and d not in pooled)
        self.pooled = pooled
        self.da = da.transpose(*self.cell_dims, *pooled, group.dim)
        v = np.asarray(self.da.values, dt)
        k = len(self.cell_dims)
        self.cell_shape = v.shape[:k]
        self.pool_shape = v.shape[k:-1]
        self.P = int(np.prod(self.pool_shape, dtype=np.int64)) if pooled else 1
        self.T = v.shape[-1]
        self.x3 = v.reshape(-1, self.P, self.T)

    def flat(self):
        """members as cells: [cell * member, time]"""
        return self.x3.reshape(-1, self.T)

    def folded(self, gidx, W):
        """the pooled series [cell, P * (T + W // 2)] and the group of each of its steps (-1 in the gaps)"""
        return _fold(self.x3, gidx, W)


def _fold(x3, gidx, W):
    C, P, T = x3.shape
    if P == 1:
        return x3.reshape(C, T), np.asarray(gidx, np.int32)
    h = W // 2
    out = np.full((C, P, T + h), np.nan, x3.dtype)
    out[:, :, :T] = x3
    g = np.full((P, T + h), -1, np.int32)
    g[:, :T] = gidx
    return out.reshape(C, -1), g.reshape(-1)


def _unfold(y2, P, T):
    """inverse of _fold for a result series: [cell, P * (T + h)] -> [cell * member, T]"""
    C = y2.shape[0]
    if P == 1:
        return y2
    return np.ascontiguousarray(y2.reshape(C, P, -1)[:, :, :T]).reshape(C * P, T)


def _common_dtype(*arrays):
    dt = max((np.dtype(getattr(a, "dtype", None) or np.asarray(a).dtype) for a in arrays), key=lambda d: d.itemsize)  # base.py:681-685
    return np.result_type(dt, np.float32)


def _train_dataset(ref: DataArray, group: Grouper, quantiles, cell_dims, cell_shape, **tables):
    """Dataset with the reference's variables/dims/coords: (cells..., <prop>, quantiles) (base.py:653-694)."""
    prop = group.prop
    crd = group.get_coordinate(ref)
    G, Q = crd.size, len(quantiles)
    coords = {k: c for k, c in ref.coords.items() if k in cell_dims}
    coords[prop] = Coordinate(prop, crd.values)
    coords["quantiles"] = Coordinate("quantiles", np.asarray(quantiles))
    out = {}
    members = tables.pop("_members", None)  # (dims, shape, coords) of the `add_dims` that P0_ref / P0_hist / pth keep
    for name, tab in tables.items():
        if members is not None and name in ("P0_ref", "P0_hist", "pth"):
            mdims, mshape, mcoords = members
            c2 = {k: c for k, c in coords.items() if k != "quantiles"}
            c2.update(mcoords)
            out[name] = DataArray(np.asarray(tab).reshape(cell_shape + tuple(mshape) + (G,)), cell_dims + tuple(mdims) + (prop,), c2)
            continue
        if getattr(tab, "strides", None) is not None and tab.size and not any(tab.strides):  # a zero-stride placeholder stays one
            tab = _Placeholder(tab)
        if isinstance(tab, _Placeholder):
            full = cell_shape + tuple(tab.shape[1:])
            dims = cell_dims + ((prop, "quantiles") if len(tab.shape) == 3 else (prop,))
            c2 = coords if len(tab.shape) == 3 else {k: c for k, c in coords.items() if k != "quantiles"}
            out[name] = DataArray(np.broadcast_to(tab.value, full), dims, c2)
            continue
        if tab.ndim == 3:
            out[name] = DataArray(tab.reshape(cell_shape + (G, Q)), cell_dims + (prop, "quantiles"), coords)
        else:
            c2 = {k: c for k, c in coords.items() if k != "quantiles"}
            out[name] = DataArray(tab.reshape(cell_shape + (G,)), cell_dims + (prop,), c2)
    return Dataset(out)


class _Placeholder:
    """a constant table of a given shape (no memory behind it)"""

    def __init__(self, arr):
        self.shape, self.value = tuple(arr.shape), np.asarray(arr.reshape(-1)[0] if arr.size else np.nan, arr.dtype)


def _nan_like(tab):
    """the all-NaN placeholder of hist_q_raw (:273-275) as a read-only zero-stride array: no memory behind it, and a device
    table is not downloaded to build it"""
    return np.broadcast_to(np.asarray(np.nan, tab.dtype), tuple(tab.shape))


@dataclass
class _TrainBlock:
    """ref and hist of one training call as [cell, time'] series blocks (pooled members folded into time')"""

    r2: object = None        # ref block ([cell, time'] host array or device view)
    h2: object = None        # hist block
    gidx_r: object = None    # group of every step of ref's (folded) time axis
    gidx_h: object = None    # ... of hist's
    same: bool = True        # ref and hist share the time axis: one fused training call
    mr: object = None        # add_dims: the [cell, member, time] views of ref and hist (_Members), the plain group index
    mh: object = None
    gidx_plain: object = None


def _prep_train(ds, group):
    _check_group(group, ds)
    ref, hist = as_dataarray(ds["ref"]), as_dataarray(ds["hist"])
    dt = _common_dtype(ref, hist)
    gidx, G = group.group_index(ref)
    tb = _TrainBlock()
    if not group.add_dims:
        tb.r2, tb.h2, cell_dims, cell_shape = cell_time_pair(ref.astype(dt) if ref.dtype != dt else ref,
                                                             hist.astype(dt) if hist.dtype != dt else hist, group.dim)
        tb.gidx_r = tb.gidx_h = gidx
        tb.same = True
        return ref, tb, gidx, G, cell_dims, cell_shape, dt
    mr, mh = _Members(ref, group, dt), _Members(hist, group, dt)
    if set(mr.cell_dims) != set(mh.cell_dims):
        raise ValueError(f"ref and hist must share their non-pooled dimensions, got {mr.cell_dims} and {mh.cell_dims}")
    if mh.cell_dims != mr.cell_dims:
        mh = _Members(hist.transpose(*mr.cell_dims, *mh.pooled, group.dim), group, dt)
    tb.r2, tb.gidx_r = mr.folded(gidx, group.window)
    tb.h2, tb.gidx_h = mh.folded(gidx, group.window)
    tb.same = mr.P == mh.P  # one fused training call needs ref and hist on the same (folded) time axis
    tb.mr, tb.mh, tb.gidx_plain = mr, mh, gidx
    return mr.da, tb, tb.gidx_h, G, mr.cell_dims, mr.cell_shape, dt


def _corr(x, y, kind):
    """utils.get_correction (:130-146) on trained tables (not series): the factor that takes x to y"""
    with np.errstate(divide="ignore", invalid="ignore"):
        return (y - x if kind == u.ADDITIVE else y / x).astype(x.dtype, copy=False)


def _preprocess_hist(ref, tb, G, group, adapt_freq_thresh, jitter_under_thresh_value, jitter_over_thresh_value,
                     jitter_over_thresh_upper_bnd):
    """_adjustment._preprocess_dataset (:48-83) on the [cell, time] block of hist: jitter under / over, then the
    frequency adaptation against ref over the windowed group blocks.  Returns (hist block, extra tables or None)."""
    from .processing import _draw_seed, quantity_to_float

    if (jitter_over_thresh_value is None) ^ (jitter_over_thresh_upper_bnd is None):
        raise ValueError("`jitter_over_thresh_value` and `jitter_over_thresh_upper_bnd` must both be specified or both be `None` (default)")
    h2 = tb.h2
    if jitter_under_thresh_value:
        h2 = hip.jitter(h2, lower=quantity_to_float(jitter_under_thresh_value, ref, "jitter_under_thresh_value"), seed=_draw_seed())
    if jitter_over_thresh_value:
        h2 = hip.jitter(h2, upper=quantity_to_float(jitter_over_thresh_value, ref, "jitter_over_thresh_value"),
                        maximum=quantity_to_float(jitter_over_thresh_upper_bnd, ref, "jitter_over_thresh_upper_bnd"), seed=_draw_seed())
    extra = None
    if adapt_freq_thresh:
        th = quantity_to_float(adapt_freq_thresh, ref, "adapt_freq_thresh")
        if tb.mh is None:
            h2, pth, _, P0_ref, P0_hist = hip.adapt_freq(h2, tb.r2, tb.gidx_h, G, group.window, th, seed=_draw_seed())
            extra = {"P0_ref": P0_ref.astype(h2.dtype), "P0_hist": P0_hist.astype(h2.dtype), "pth": pth}
        else:
            # `add_dims`: the members are NOT pooled for the frequency adaptation (_processing.py:85-86; P0_ref / P0_hist / pth
            # keep the extra dimensions, _adjustment.py:86-95) — every member of hist is adapted on its own against its member
            # of ref, a dataset without the dimension broadcast along it (_processing.py:104-106) — and the adapted members are
            # pooled afterwards for the quantiles.
            mr, mh = tb.mr, tb.mh
            P = max(mr.P, mh.P)
            if mr.P not in (1, P) or mh.P not in (1, P) or (mr.P == mh.P and mr.pool_shape != mh.pool_shape):
                raise NotImplementedError("adapt_freq_thresh with differently shaped `add_dims` dimensions in ref and hist is not implemented on the HIP path")
            C, T = mh.x3.shape[0], mh.T
            dt = mh.x3.dtype
            hx = np.asarray(_unfold(np.asarray(h2), mh.P, T)).reshape(C, mh.P, T)  # (hist after the jitter steps, member by member)
            hb = np.ascontiguousarray(np.broadcast_to(hx, (C, P, T))).reshape(C * P, T)
            rb = np.ascontiguousarray(np.broadcast_to(mr.x3, (C, P, T))).reshape(C * P, T)
            h_ad, pth, _, P0_ref, P0_hist = hip.adapt_freq(hb, rb, tb.gidx_plain, G, group.window, th, seed=_draw_seed())
            h2, tb.gidx_h = _fold(np.asarray(h_ad).reshape(C, P, T), tb.gidx_plain, group.window)
            if mh.P != P:
                tb.same = False  # (hist gained members: it no longer shares ref's folded axis, nor the group index the caller holds)
            src = mh if mh.P == P else mr
            members = (src.pooled, src.pool_shape, {d: src.da.coords[d] for d in src.pooled if d in src.da.coords})
            extra = {"P0_ref": np.asarray(P0_ref).astype(dt), "P0_hist": np.asarray(P0_hist).astype(dt), "pth": np.asarray(pth), "_members": members}
    return h2, extra


def _fused_eqm_train(ds, *, group, kind, quantiles, adapt_freq_thresh=None, jitter_under_thresh_value=None,
                     jitter_over_thresh_value=None, jitter_over_thresh_upper_bnd=None, max_tail_factor=None):
    ref, tb, gidx, G, cell_dims, cell_shape, dt = _prep_train(ds, group)
    q = np.asarray(quantiles, dt)
    h_raw = tb.h2
    h2, extra = _preprocess_hist(ref, tb, G, group, adapt_freq_thresh, jitter_under_thresh_value,
                                 jitter_over_thresh_value, jitter_over_thresh_upper_bnd)
    if tb.same:
        af, hq = hip.eqm_train(tb.r2, h2, gidx, G, group.window, q, kind)
    else:  # differently pooled ref / hist: each on its own time axis, the factors from the two tables
        hq = hip.quantile_groups(h2, tb.gidx_h, G, group.window, q)
        af = _corr(hq, hip.quantile_groups(tb.r2, tb.gidx_r, G, group.window, q), kind)
    # hist_q_raw = quantiles of hist BEFORE the pre-processing steps (:254-256); a NaN dummy unless max_tail_factor (:273-275)
    if max_tail_factor is None:
        hq_raw = _nan_like(hq)
    else:
        hq_raw = hq.copy() if h2 is h_raw else hip.quantile_groups(h_raw, gidx, G, group.window, q)
    out = _train_dataset(ref, group, quantiles, cell_dims, cell_shape, af=af, hist_q=hq, hist_q_raw=hq_raw, **(extra or {}))
    out["af"].attrs["kind"] = kind
    return out


def _fused_dqm_train(ds, *, group, kind, quantiles, adapt_freq_thresh=None, jitter_under_thresh_value=None,
                     jitter_over_thresh_value=None, jitter_over_thresh_upper_bnd=None, max_tail_factor=None):
    ref, tb, gidx, G, cell_dims, cell_shape, dt = _prep_train(ds, group)
    q = np.asarray(quantiles, dt)
    h_raw = tb.h2
    h2, extra = _preprocess_hist(ref, tb, G, group, adapt_freq_thresh, jitter_under_thresh_value,
                                 jitter_over_thresh_value, jitter_over_thresh_upper_bnd)
    if tb.same:
        af, hq, sc = hip.dqm_train(tb.r2, h2, gidx, G, group.window, q, kind)
    else:  # differently pooled ref / hist: normalised quantiles and group means of each on its own time axis
        _, hq, _ = hip.dqm_train(h2, h2, tb.gidx_h, G, group.window, q, kind)
        _, rq, _ = hip.dqm_train(tb.r2, tb.r2, tb.gidx_r, G, group.window, q, kind)
        af = _corr(hq, rq, kind)
        sc = _corr(hip.group_means(h2, tb.gidx_h, G, group.window).astype(dt, copy=False),
                   hip.group_means(tb.r2, tb.gidx_r, G, group.window).astype(dt, copy=False), kind)
    # hist_q_raw: quantiles of the raw (not pre-processed, not normalised) hist (:151-153)
    hq_raw = hip.quantile_groups(h_raw, gidx, G, group.window, q) if max_tail_factor is not None else _nan_like(hq)
    out = _train_dataset(ref, group, quantiles, cell_dims, cell_shape, af=af, hist_q=hq, hist_q_raw=hq_raw, scaling=sc,
                         **(extra or {}))
    out["af"].attrs["kind"] = kind
    out["scaling"].attrs["kind"] = kind
    return out


@map_groups(
    fused=_fused_dqm_train,
    af=[Grouper.PROP, "quantiles"],
    hist_q=[Grouper.PROP, "quantiles"],
    hist_q_raw=[Grouper.PROP, "quantiles"],
    scaling=[Grouper.PROP],
)
def dqm_train(ds, *, dim, kind, quantiles, adapt_freq_thresh=None, jitter_under_thresh_value=None,
              jitter_over_thresh_value=None, jitter_over_thresh_upper_bnd=None, max_tail_factor=None) -> Dataset:
    """Train step on one group (the per-group body kept for ``Grouper.apply(dqm_train.func, ...)``;
    _adjustment.py:150-190): group-mean normalisation, quantiles, af, scaling."""
    _unsupported(adapt_freq_thresh=adapt_freq_thresh, jitter_under_thresh_value=jitter_under_thresh_value,
                 jitter_over_thresh_value=jitter_over_thresh_value, jitter_over_thresh_upper_bnd=jitter_over_thresh_upper_bnd)
    ref_dim = Grouper.filter_dim(ds.ref, dim)
    sim_dim = Grouper.filter_dim(ds.hist, dim)
    hist_q_raw = nbu.quantile(ds.hist, quantiles, sim_dim) if max_tail_factor is not None else None
    mu_ref, mu_hist = ds.ref.mean(ref_dim), ds.hist.mean(sim_dim)
    refn = u.apply_correction(ds.ref, u.invert(mu_ref, kind), kind)
    histn = u.apply_correction(ds.hist, u.invert(mu_hist, kind), kind)
    ref_q = nbu.quantile(refn, quantiles, ref_dim)
    hist_q = nbu.quantile(histn, quantiles, sim_dim)
    af = u.get_correction(hist_q, ref_q, kind)
    scaling = u.get_correction(mu_hist, mu_ref, kind=kind)
    if hist_q_raw is None:
        hist_q_raw = full_like(hist_q, np.nan)
    return Dataset({"af": af, "hist_q": hist_q, "hist_q_raw": hist_q_raw, "scaling": scaling})


@map_groups(
    fused=_fused_eqm_train,
    af=[Grouper.PROP, "quantiles"],
    hist_q=[Grouper.PROP, "quantiles"],
    hist_q_raw=[Grouper.PROP, "quantiles"],
)
def eqm_train(ds, *, dim, kind, quantiles, adapt_freq_thresh=None, jitter_under_thresh_value=None,
              jitter_over_thresh_value=None, jitter_over_thresh_upper_bnd=None, max_tail_factor=None) -> Dataset:
    """EQM: train step on one group (per-group body, _adjustment.py:253-286)."""
    _unsupported(adapt_freq_thresh=adapt_freq_thresh, jitter_under_thresh_value=jitter_under_thresh_value,
                 jitter_over_thresh_value=jitter_over_thresh_value, jitter_over_thresh_upper_bnd=jitter_over_thresh_upper_bnd)
    ref_q = nbu.quantile(ds.ref, quantiles, Grouper.filter_dim(ds.ref, dim))
    hist_q = nbu.quantile(ds.hist, quantiles, Grouper.filter_dim(ds.hist, dim))
    af = u.get_correction(hist_q, ref_q, kind)
    hist_q_raw = hist_q.copy() if max_tail_factor is not None else full_like(hist_q, np.nan)
    return Dataset({"af": af, "hist_q": hist_q, "hist_q_raw": hist_q_raw})


def _prep_adjust(ds, group: Grouper, interp: str, grouped_linear: bool = False):
    _check_group(group, ds)
    if interp not in ("nearest", "linear", "cubic"):
        raise ValueError(f"interp must be 'nearest', 'linear' or 'cubic', got {interp!r}")
    if interp != "nearest" and group.prop != "group" and not grouped_linear:
        raise NotImplementedError(f"interp={interp!r} with group={group.name!r} is not implemented on the HIP path")
    sim = as_dataarray(ds["sim"])
    if group.add_dims and "P0_ref" in ds.data_vars:
        # the trained frequency tables keep the `add_dims` (reference _adjustment.py:86-95): a sim without them is broadcast
        # along them, as xarray does when the reference combines the two (_processing.py:92-96)
        from .processing import _broadcast_along

        sim = _broadcast_along(sim, as_dataarray(ds["P0_ref"]), list(group.add_dims))
    dt = _common_dtype(sim, as_dataarray(ds["af"]))
    pooled = [d for d in group.add_dims if d in sim.dims]
    if pooled:  # the pooled members become cells of their own, next to each other (the tables are broadcast over them)
        sim = sim.transpose(*[d for d in sim.dims if d != group.dim and d not in pooled], *pooled, group.dim)
    s2, cell_dims, cell_shape = cell_time_view(sim.astype(dt) if sim.dtype != dt else sim, group.dim)
    gidx, G = group.group_index(sim)
    return sim, s2, gidx, G, cell_dims, cell_shape, dt


def _linear_group_coords(group: Grouper, sim: DataArray):
    """What utils.interp_on_quantiles (utils.py:476-489) hands to griddata next to the values: the coordinates of the table
    rows wrapped with add_cyclic_bounds(cyclic_coords=False) (:284-314; seasons as 0..3, :403-405) and the fractional group
    coordinate of every time step, Grouper.get_index(interp=True) (base.py:308-320)."""
    crd = np.asarray(group.get_coordinate(sim).values)
    if group.prop == "season":
        crd = np.arange(4)
    crd = crd.astype(np.float64)
    gcoord = np.concatenate([[crd[0] - (crd[1] - crd[0])], crd, [crd[-1] + (crd[-1] - crd[-2])]])
    newg = np.asarray(group.get_index(sim, interp=True).values, np.float64)
    return gcoord, newg


def _n_pooled(group: Grouper, cell_dims, cell_shape):
    """number of pooled members per cell in the [cell * member, time] block of _prep_adjust"""
    return int(np.prod([n for d, n in zip(cell_dims, cell_shape) if d in group.add_dims], dtype=np.int64))


def _adapt_freq_sim(ds, sim, s2, gidx, G, cell_dims, group, dt, adapt_freq_thresh, cell_shape=None):
    """_adapt_freq_preprocess on the adjust side (_adjustment.py:32-46, 639-645): the frequency adaptation of sim with the
    trained P0_ref / P0_hist / pth, on the groups WITHOUT window (Grouper(group.name))."""
    if not adapt_freq_thresh:
        return s2
    from .processing import _draw_seed, quantity_to_float

    th = quantity_to_float(adapt_freq_thresh, sim, "adapt_freq_thresh")
    tabs = [table_view(as_dataarray(ds[k]), cell_dims, group.prop, last=None, cell_shape=cell_shape) for k in ("P0_ref", "P0_hist", "pth")]
    out = hip.adapt_freq(s2, None, gidx, G, 1, th, seed=_draw_seed(), P0_ref=tabs[0].astype(np.float64),
                         P0_hist=tabs[1].astype(np.float64), pth=tabs[2].astype(dt, copy=False))
    return out[0]


def _tail_mask(ds, s2, scen, gidx, G, cell_dims, group, dt, max_tail_factor, cell_shape=None, sim=None, interp="nearest"):
    """mask no bias adjustment where sim is larger than n times the largest raw hist quantile (_adjustment.py:648-658,
    672-673).  The last raw quantile is broadcast on the time steps like the reference does (:650-657): the value of the
    step's group ("nearest", and always on day-of-year groups), or interpolated between the group centres with ``interp``
    — then the broadcast table is formed as a series on the device (0 + factor) and the mask runs with one "group" per
    time step."""
    if max_tail_factor is None:
        return scen
    hq_raw = table_view(as_dataarray(ds["hist_q_raw"]), cell_dims, group.prop, cell_shape=cell_shape).astype(dt, copy=False)
    if interp == "nearest" or G == 1 or group.prop == "dayofyear":
        return hip.tail_mask(s2, np.ascontiguousarray(scen), hq_raw, gidx if G > 1 else None, max_tail_factor)
    Cn, Tn = s2.shape
    last = np.ascontiguousarray(hq_raw[:, :, -1])
    thr = _broadcast_correct(group, sim, as_dataarray(ds["hist_q_raw"]), np.zeros((Cn, Tn), dt), last, gidx, G, "+", interp)
    return hip.tail_mask(s2, np.ascontiguousarray(scen), np.ascontiguousarray(thr).reshape(Cn, Tn, 1),
                         np.arange(Tn, dtype=np.int32), max_tail_factor)


def _series_like(sim: DataArray, arr, cell_dims, cell_shape, dim, name, attrs=None):
    coords = {k: c for k, c in sim.coords.items() if k in cell_dims or k == dim}
    return DataArray(arr.reshape(cell_shape + (arr.shape[-1],)), cell_dims + (dim,), coords, attrs or {}, name)


@map_blocks(reduces=[Grouper.PROP, "quantiles"], scen=[])
def qm_adjust(ds, *, group, interp, extrapolation, kind, adapt_freq_thresh=None, max_tail_factor=None) -> Dataset:
    """QM (DQM and EQM): adjust step on one block (_adjustment.py:594-676)."""
    sim, s2, gidx, G, cell_dims, cell_shape, dt = _prep_adjust(ds, group, interp, grouped_linear=True)
    s2 = _adapt_freq_sim(ds, sim, s2, gidx, G, cell_dims, group, dt, adapt_freq_thresh, cell_shape)
    af = table_view(as_dataarray(ds["af"]), cell_dims, group.prop, cell_shape=cell_shape).astype(dt, copy=False)
    hq = table_view(as_dataarray(ds["hist_q"]), cell_dims, group.prop, cell_shape=cell_shape).astype(dt, copy=False)
    if interp in ("linear", "cubic") and G > 1:
        # the reference's griddata path (utils.py:391-396): "linear" = piecewise-linear on the Delaunay triangulation of the
        # cell's table nodes, "cubic" = the Clough-Tocher interpolant on the same triangulation (csrc/lin2d.hpp)
        gcoord, newg = _linear_group_coords(group, sim)
        scen = hip.qm_adjust_linear2d(s2, af, hq, gcoord, newg, extrapolation, kind, cubic=interp == "cubic")
    elif G > 1 and max_tail_factor is None and getattr(s2, "cell_time_strides", None) is not None:
        # a (time, cells...) series resident on the device: mapped in place by the lane-per-cell kernel (csrc/adjust_tm.hpp) and
        # handed back in the same layout — what the reference returns for a (time, lat, lon) sim, without a layout pass
        scen = hip.qm_adjust(s2, af, hq, gidx, interp, extrapolation, kind, out_time_major=True)  # [time, cell]
        coords = {k: c for k, c in sim.coords.items() if k in cell_dims or k == group.dim}
        return DataArray(scen.reshape((scen.shape[0],) + tuple(cell_shape)), (group.dim,) + tuple(cell_dims), coords, {}, "scen").to_dataset()
    else:
        scen = hip.qm_adjust(s2, af, hq, gidx if G > 1 else None, interp, extrapolation, kind)
    scen = _tail_mask(ds, s2, scen, gidx, G, cell_dims, group, dt, max_tail_factor, cell_shape, sim=sim, interp=interp)
    return _series_like(sim, scen, cell_dims, cell_shape, group.dim, "scen").to_dataset()


def _pooled_trend(s2, sc, gidx, G, group, kind, tnum, degree, P):
    """The detrending fit with pooled members (detrending.py:196-203): ONE trend per cell, fitted on the mean of the scaled
    series over the pooled members (and the window); every member is detrended / retrended with it.
    s2, sc: [cell * member, time] / [cell * member, group] blocks; returns the trend repeated per member."""
    CP, T = s2.shape
    if group.window > 1 and np.isnan(s2).any():
        raise NotImplementedError("dqm_adjust with `add_dims`, a window and missing values in sim is not implemented on the HIP "
                                  "path (the pooled window mean is formed as the window mean of the member means)")
    scaled = hip.scaling_adjust(s2, sc, gidx if G > 1 else None, kind)
    # mean over the members of every (cell, time): series of P values with one group
    by_time = np.ascontiguousarray(scaled.reshape(CP // P, P, T).transpose(0, 2, 1)).reshape(-1, P)
    mean = hip.group_means(by_time, None, 1, 1).reshape(CP // P, T).astype(s2.dtype, copy=False)
    trend = hip.polydetrend_trend(mean, gidx, G, group.window, tnum, degree)
    return np.ascontiguousarray(np.repeat(trend, P, axis=0))


@map_blocks(reduces=[Grouper.PROP, "quantiles"], scen=[], trend=[])
def dqm_adjust(ds, *, group, interp, kind, extrapolation, detrend, adapt_freq_thresh=None, max_tail_factor=None, _outputs=None) -> Dataset:
    """DQM adjustment on one block (_adjustment.py:679-780): scaling, per-group linear detrending of the window
    means, quantile mapping, retrending — one trend kernel + one mapping kernel on the device.

    ``_outputs`` (extension): the declared outputs the caller will keep; None = all of them (scen, trend), like the
    reference.  `DetrendedQuantileMapping._adjust` keeps `.scen` only (adjustment.py:659-668) and says so: the mapping kernel
    then evaluates the per-group regression line itself instead of reading a trend array written for nobody."""
    own = None  # a PolyDetrend with a group / kind / options of its own: its trend is fitted first and handed to the mapping
    if isinstance(detrend, PolyDetrend):
        degree = int(detrend.degree)
        if (detrend.group.name != group.name or detrend.group.window != group.window or detrend.group.add_dims != group.add_dims
                or detrend.kind != kind or detrend.preserve_mean or detrend.mult_skip_zeros):
            own = detrend
    elif isinstance(detrend, (int, np.integer)):
        degree = int(detrend)
    elif isinstance(detrend, (MeanDetrend, NoDetrend, RollingMeanDetrend)):
        degree, own = 0, detrend
    else:
        raise NotImplementedError("dqm_adjust on the HIP path supports an int degree, PolyDetrend, MeanDetrend, RollingMeanDetrend or NoDetrend")
    if not 0 <= degree <= 6:
        raise NotImplementedError(f"PolyDetrend(degree={degree}) is not implemented on the HIP path (0..6)")
    sim, s2, gidx, G, cell_dims, cell_shape, dt = _prep_adjust(ds, group, interp, grouped_linear=True)
    s2 = _adapt_freq_sim(ds, sim, s2, gidx, G, cell_dims, group, dt, adapt_freq_thresh, cell_shape)
    af = table_view(as_dataarray(ds["af"]), cell_dims, group.prop, cell_shape=cell_shape).astype(dt, copy=False)
    hq = table_view(as_dataarray(ds["hist_q"]), cell_dims, group.prop, cell_shape=cell_shape).astype(dt, copy=False)
    sc = table_view(as_dataarray(ds["scaling"]), cell_dims, group.prop, last=None, cell_shape=cell_shape).astype(dt, copy=False)
    tnum = group._time_index(sim).numeric_ns
    P = _n_pooled(group, cell_dims, cell_shape)
    if own is not None:
        if (interp in ("linear", "cubic") and G > 1) or P > 1 or own.group.add_dims:
            raise NotImplementedError("a PolyDetrend with its own group / kind / options together with `add_dims` or a grouped "
                                      "interp='linear' is not implemented on the HIP path")
        if own.mult_skip_zeros and own.kind != "*":  # detrending.py:107-110
            warnings.warn(f"mult_skip_zeros is only used for kind='*'. Your kind is {own.kind}. mult_skip_zeros will be ignored.",
                          UserWarning, stacklevel=2)
        gidx_d, G_d = own.group.group_index(sim)
        scaled = hip.scaling_adjust(s2, sc, gidx if G > 1 else None, kind)
        own_kind = own.kind
        if isinstance(own, NoDetrend):  # nothing is removed: a zero additive trend
            trend_in, own_kind = np.zeros_like(scaled), "+"
        elif isinstance(own, MeanDetrend):
            means = hip.group_means(scaled, gidx_d, G_d, own.group.window)
            trend_in = hip.scaling_adjust(np.zeros_like(scaled), means, gidx_d if G_d > 1 else None, "+")
        elif isinstance(own, RollingMeanDetrend):
            if own.group.prop != "group":
                raise NotImplementedError("RollingMeanDetrend is implemented for group='time' on the HIP path")
            trend_in = hip.rolling_mean(scaled, int(own.win), own.min_periods)
        else:
            trend_in = hip.polydetrend_trend(scaled, gidx_d, G_d, own.group.window, tnum, degree, bool(own.preserve_mean), own.kind)
        scen, trend = hip.dqm_adjust(s2, af, hq, sc, gidx, group.window, None, degree, interp, extrapolation, kind, trend_in=trend_in,
                                     trend_kind=own_kind, mult_skip_zeros=bool(own.mult_skip_zeros) and own_kind == "*")
    elif interp in ("linear", "cubic") and G > 1:
        # the scaling is broadcast between the group centres with the same method, except on day-of-year groups (:748-757);
        # then fit / detrend / grouped-linear (or Clough-Tocher) mapping / retrend (:759-773) as separate device steps
        if P > 1:
            raise NotImplementedError("dqm_adjust with `add_dims` and a grouped interp='linear' / 'cubic' is not implemented on the HIP path")
        scaled = _broadcast_correct(group, sim, as_dataarray(ds["scaling"]), s2, sc, gidx, G, kind,
                                    interp if group.prop != "dayofyear" else "nearest")
        trend = hip.polydetrend_trend(scaled, gidx, G, group.window, tnum, degree)
        gcoord, newg = _linear_group_coords(group, sim)
        scen = hip.qm_adjust_linear2d(scaled, af, hq, gcoord, newg, extrapolation, kind, trend=trend, cubic=interp == "cubic")
    elif P > 1:
        trend_in = _pooled_trend(s2, sc, gidx, G, group, kind, tnum, degree, P)
        scen, trend = hip.dqm_adjust(s2, af, hq, sc, gidx, group.window, None, degree, interp, extrapolation, kind, trend_in=trend_in)
    else:
        scen, trend = hip.dqm_adjust(s2, af, hq, sc, gidx, group.window, tnum, degree, interp, extrapolation, kind,
                                     want_trend=_outputs is None or "trend" in _outputs)
    scen = _tail_mask(ds, s2, scen, gidx, G, cell_dims, group, dt, max_tail_factor, cell_shape, sim=sim, interp=interp)
    out = {"scen": _series_like(sim, scen, cell_dims, cell_shape, group.dim, "scen")}
    if trend is not None:
        out["trend"] = _series_like(sim, trend, cell_dims, cell_shape, group.dim, "trend")
    return Dataset(out)


@map_blocks(reduces=[Grouper.PROP, "quantiles"], scen=[], sim_q=[])
def qdm_adjust(ds, *, group, interp, extrapolation, kind, adapt_freq_thresh=None, rank_window=None, max_tail_factor=None,
               _outputs=None) -> Dataset:
    """QDM adjustment on one block (_adjustment.py:783-886).  ``_outputs`` (extension): the declared outputs the caller will
    keep; None = all of them (scen, sim_q), like the reference."""
    if rank_window is None:
        rank_window = False
        if group.window > 1:
            warnings.warn(
                "QDM method can now perform the adjustment step by expanding the time dimension "
                "with the same window as used in the training. This can already be used by setting "
                "`rank_window = True`. This will be the only possible behaviour in `xsdba>=0.8`.  "
                "The current behaviour is obtained by setting `rank_window = False` and will be "
                "deprecated in `xsdba>=0.8`. It will still be possible to use the old behaviour by "
                "monkeypatching the group argument in the QDM class between the training and "
                "adjustment, though this behaviour is not recommended.",
                category=DeprecationWarning,
                stacklevel=2,
            )
    sim, s2, gidx, G, cell_dims, cell_shape, dt = _prep_adjust(ds, group, interp, grouped_linear=True)
    s2 = _adapt_freq_sim(ds, sim, s2, gidx, G, cell_dims, group, dt, adapt_freq_thresh, cell_shape)
    af = table_view(as_dataarray(ds["af"]), cell_dims, group.prop, cell_shape=cell_shape).astype(dt, copy=False)
    quantiles = np.asarray(ds["quantiles"].values, dt)
    P = _n_pooled(group, cell_dims, cell_shape)
    sim_q = None
    if interp in ("linear", "cubic") and G > 1:
        # the factors interpolated at (sim_q, fractional group) over the regular (quantile, group) grid (:873-881).  Every
        # cell of that grid is a cocircular quadruple: which diagonal splits it is the triangulator's choice (Qhull's in
        # the reference, the sweep's here), so the two agree exactly only where the four corner factors are coplanar.
        if P > 1 and rank_window:
            raise NotImplementedError("qdm_adjust with `add_dims`, rank_window and a grouped interp='linear' / 'cubic' is not implemented on the HIP path")
        sim_q = hip.rank_groups(s2, gidx, G, group.window, bool(rank_window))
        gcoord, newg = _linear_group_coords(group, sim)
        scen = hip.qm_adjust_linear2d(s2, af, quantiles, gcoord, newg, extrapolation, kind, lookup=sim_q, cubic=interp == "cubic")
    elif P > 1 and rank_window:
        # ranks over (time, window, pooled members) (_adjustment.py:873: main_only = not rank_window): the members of a
        # cell laid end to end as one series, like at training; the table of the cell serves all of them
        T = s2.shape[1]
        f2, gf = _fold(s2.reshape(-1, P, T), gidx, group.window)
        scen, sim_q = hip.qdm_adjust(f2, np.ascontiguousarray(af[::P]), quantiles, gf, group.window, True, interp, extrapolation, kind)
        scen, sim_q = _unfold(scen, P, T), _unfold(sim_q, P, T)
    else:
        # sim_q (a float64 series as large as two copies of sim) is only produced when the caller keeps it
        scen, sim_q = hip.qdm_adjust(s2, af, quantiles, gidx, group.window, bool(rank_window), interp, extrapolation, kind,
                                     want_sim_q=_outputs is None or "sim_q" in _outputs)
    scen = _tail_mask(ds, s2, scen, gidx, G, cell_dims, group, dt, max_tail_factor, cell_shape, sim=sim, interp=interp)
    out = {"scen": _series_like(sim, scen, cell_dims, cell_shape, group.dim, "scen")}
    if sim_q is not None:
        out["sim_q"] = _series_like(sim, sim_q, cell_dims, cell_shape, group.dim, "sim_q")
    return Dataset(out)


def _fused_scaling_train(ds, *, group, kind):
    ref, tb, gidx, G, cell_dims, cell_shape, dt = _prep_train(ds, group)
    mhist = hip.group_means(tb.h2, tb.gidx_h, G, group.window)
    mref = hip.group_means(tb.r2, tb.gidx_r, G, group.window)
    af = mref - mhist if kind == u.ADDITIVE else mref / mhist  # utils.get_correction(mhist, mref, kind) :130-146
    out = _train_dataset(ref, group, [], cell_dims, cell_shape, af=af.astype(dt, copy=False))
    out = Dataset({"af": out["af"]})
    out["af"].attrs["kind"] = kind
    return out


@map_groups(fused=_fused_scaling_train, af=[Grouper.PROP])
def scaling_train(ds, *, dim, kind) -> Dataset:
    """Scaling: train on one group (_adjustment.py:941-958): the correction between the group means of hist and ref."""
    mhist = ds.hist.mean(Grouper.filter_dim(ds.hist, dim))
    mref = ds.ref.mean(Grouper.filter_dim(ds.ref, dim))
    return Dataset({"af": u.get_correction(mhist, mref, kind)})


def _group_interp_nodes(group: Grouper, sim: DataArray, gidx, G: int, interp: str):
    """What utils.broadcast (utils.py:222-246) needs per time step to interpolate a grouped table linearly: the lower and
    upper neighbouring groups (0-based, cyclic: add_cyclic_bounds :284-314) and the fraction of the way between them, from
    the fractional group index of Grouper.get_index(interp=True) (base.py:308-320).  scipy's interp1d brackets a point that
    sits exactly on a node k with (k-1, k], hence ceil.  Returns (g_lo, g_hi, frac) or (gidx, None, None) for 'nearest' / one group."""
    if interp == "nearest" or G == 1:
        return gidx, None, None
    if interp != "linear":
        raise NotImplementedError(f"interp={interp!r} between group centres is not implemented on the HIP path ('nearest' or 'linear')")
    i = np.asarray(group.get_index(sim, interp=True).values, np.float64)
    c0 = 0 if group.prop == "season" else 1  # coordinate of the first group
    hi = np.ceil(i)
    lo = hi - 1
    frac = i - lo
    g_lo = np.mod(lo - c0, G).astype(np.int32)
    g_hi = np.mod(hi - c0, G).astype(np.int32)
    g_lo[gidx < 0] = -1
    return g_lo, g_hi, frac


def _cubic_group_series(group: Grouper, sim: DataArray, tab, G: int):
    """the [1, G] table of a lone series interpolated on its time steps like xarray's .interp(method="cubic") does inside
    utils.broadcast (utils.py:222-246): scipy's interp1d(kind="cubic") through the G + 2 values of the table wrapped with
    its cyclic rows, evaluated at Grouper.get_index(interp=True) -> [T] in the dtype of the table"""
    from scipy.interpolate import interp1d

    i = np.asarray(group.get_index(sim, interp=True).values, np.float64)
    c0 = 0 if group.prop == "season" else 1  # coordinate of the first group
    row = np.asarray(tab, np.float64).reshape(G)
    ext = np.concatenate([row[-1:], row, row[:1]])
    return interp1d(np.arange(c0 - 1, c0 + G + 1, dtype=np.float64), ext, kind="cubic", bounds_error=False)(i).astype(np.asarray(tab).dtype)


def _broadcast_correct(group: Grouper, sim: DataArray, table: DataArray, s2, tab, gidx, G: int, kind: str, interp: str):
    """apply_correction(sim, utils.broadcast(table, sim, group=group, interp=interp), kind) (utils.py:180-248) -> [C, T].
    'nearest' / 'linear' between the group centres run in ``hip.scaling_adjust``.  'cubic': the reference falls back to
    'linear' with a warning as soon as the table shares a dimension with the series (:228-234); for a lone series
    xarray's ``.interp(method="cubic")`` is scipy's interp1d(kind="cubic") through the G + 2 values of the table wrapped with
    its cyclic rows (add_cyclic_bounds :284-314) — a table-sized spline evaluated once per time step here, applied to the
    series on the device."""
    if interp == "cubic" and G > 1 and set(table.dims) & set(sim.dims):
        warnings.warn("Broadcasting operations in multiple dimensions can only be done with linear and nearest-neighbor"
                      " interpolation, not cubic. Using linear.", stacklevel=2)
        interp = "linear"
    if interp == "cubic" and G > 1:
        return hip.series_correction(s2, _cubic_group_series(group, sim, tab, G), kind)
    g_lo, g_hi, frac = _group_interp_nodes(group, sim, gidx, G, interp)
    return hip.scaling_adjust(s2, tab, g_lo if G > 1 else None, kind, g_hi=g_hi, frac=frac)


@map_blocks(reduces=[Grouper.PROP], scen=[])
def scaling_adjust(ds, *, group, interp, kind) -> Dataset:
    """Scaling: adjust on one block (_adjustment.py:961-974); the factors are broadcast on the group of each time step
    (utils.broadcast :180-248), "nearest" or linearly interpolated between group centres."""
    sim, s2, gidx, G, cell_dims, cell_shape, dt = _prep_adjust(ds, group, "nearest")
    af = table_view(as_dataarray(ds["af"]), cell_dims, group.prop, last=None, cell_shape=cell_shape).astype(dt, copy=False)
    scen = _broadcast_correct(group, sim, as_dataarray(ds["af"]), s2, af, gidx, G, kind, interp)
    return _series_like(sim, scen, cell_dims, cell_shape, group.dim, "scen").to_dataset()


@map_blocks(reduces=[Grouper.ADD_DIMS, Grouper.DIM], af=[Grouper.PROP], hist_thresh=[Grouper.PROP])
def loci_train(ds, *, group, thresh) -> Dataset:
    """LOCI: train on one block (_adjustment.py:889-919), every group at once on the device."""
    ref, tb, gidx, G, cell_dims, cell_shape, dt = _prep_train(ds, group)
    if not tb.same:
        raise NotImplementedError("LOCI with `add_dims` present in only one of ref / hist is not implemented on the HIP path")
    af, sth = hip.loci_train(tb.r2, tb.h2, gidx, G, group.window, thresh)
    out = _train_dataset(ref, group, [], cell_dims, cell_shape, af=af, hist_thresh=sth)
    return Dataset({"af": out["af"], "hist_thresh": out["hist_thresh"]})


@map_blocks(reduces=[Grouper.PROP], scen=[])
def loci_adjust(ds, *, group, thresh, interp) -> Dataset:
    """LOCI: adjust on one block (_adjustment.py:922-939); factors and thresholds of the group of each time step, "nearest"
    or linearly interpolated between group centres (the reference's default)."""
    sim, s2, gidx, G, cell_dims, cell_shape, dt = _prep_adjust(ds, group, "nearest")
    if interp == "cubic" and G > 1 and set(as_dataarray(ds["af"]).dims) & set(sim.dims):
        # utils.broadcast (utils.py:228-234): tables sharing a dimension with the series are interpolated linearly
        warnings.warn("Broadcasting operations in multiple dimensions can only be done with linear and nearest-neighbor"
                      " interpolation, not cubic. Using linear.", stacklevel=2)
        interp = "linear"
    af = table_view(as_dataarray(ds["af"]), cell_dims, group.prop, last=None, cell_shape=cell_shape).astype(dt, copy=False)
    sth = table_view(as_dataarray(ds["hist_thresh"]), cell_dims, group.prop, last=None, cell_shape=cell_shape).astype(dt, copy=False)
    if interp == "cubic" and G > 1:
        # a lone series: factor and threshold of every time step from the cubic spline through the wrapped tables, handed to
        # the kernel as tables with one "group" per time step
        Tn = s2.shape[1]
        scen = hip.loci_adjust(s2, _cubic_group_series(group, sim, af, G).reshape(1, Tn), _cubic_group_series(group, sim, sth, G).reshape(1, Tn),
                               np.arange(Tn, dtype=np.int32), thresh)
        return _series_like(sim, scen, cell_dims, cell_shape, group.dim, "scen").to_dataset()
    g_lo, g_hi, frac = _group_interp_nodes(group, sim, gidx, G, interp)
    scen = hip.loci_adjust(s2, af, sth, g_lo if G > 1 else None, thresh, g_hi=g_hi, frac=frac)
    return _series_like(sim, scen, cell_dims, cell_shape, group.dim, "scen").to_dataset()
