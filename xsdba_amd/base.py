"""Grouping objects and the map_blocks / map_groups decorator runtime of the hot path.

Mirrors the interface of the reference's ``xsdba.base`` (base.py:26-100 Parametrizable, :118-501 Grouper,
:504-538 parse_group, :563-778 map_blocks / map_groups) for the quantile-mapping path: same names,
arguments, attributes written on the outputs and error behaviour.  What differs is the engine: the
reference builds ``rolling(...).construct`` windows and loops over ``groupby`` groups in Python; here a
``Grouper`` only produces the per-time group index (+ window) that the HIP kernels consume, and
``map_groups`` functions on the path carry a *fused* device implementation that handles every group of
every cell in one launch.  ``Grouper.apply`` with a user function still works (host loop over groups on
numpy data) so third-party ``map_groups`` functions keep running — that generic loop is host plumbing,
not a fallback of the accelerated path.
"""
from __future__ import annotations

import json
from collections import OrderedDict, UserDict
from inspect import _empty, signature

import numpy as np

from .calendar import MAX_DOY, TimeIndex
from .labeled import Coordinate, DataArray, Dataset, as_dataarray, as_dataset


class Parametrizable(UserDict):
    """A mapping that IS its parameters: attribute access reads them, the repr shows the non-default ones and pickling keeps
    nothing else (the reference's contract, base.py:26-72)."""

    _repr_hide_params: list = []

    def __getstate__(self):
        return self.data

    def __setstate__(self, state):
        self.data = dict(state)

    def __getattr__(self, attr):
        # only reached when normal lookup failed; "data" itself may be missing while unpickling
        store = self.__dict__.get("data")
        if attr != "data" and store is not None and attr in store:
            return store[attr]
        return self.__getattribute__(attr)

    @property
    def parameters(self) -> dict:
        return dict(self.data)

    def _is_default(self, name, value) -> bool:
        par = signature(self.__init__).parameters.get(name)
        if par is None or par.default is _empty:
            return False
        if par.default is None:  # None stands for "nothing given": empty containers count as default too
            return value in ([], {}, set(), None)
        return value in [par.default]

    def __repr__(self) -> str:
        shown = [f"{k}={v!r}" for k, v in self.items() if k not in self._repr_hide_params and not self._is_default(k, v)]
        return f"{type(self).__name__}({', '.join(shown)})"


# ---- jsonpickle-compatible encoding of Parametrizable objects (base.py:80-100) -------------------------
# jsonpickle is not installed on the GPU image, so the small subset of its format these objects need is
# written and parsed by hand: {"py/object": "<module>.<Class>", "py/state": {...}} with nested Grouper.
_REGISTRY: dict[str, type] = {}


def _register(cls, ref_module: str):
    cls._ref_qualname = f"{ref_module}.{cls.__name__}"
    _REGISTRY[cls._ref_qualname] = cls
    _REGISTRY[f"{cls.__module__}.{cls.__name__}"] = cls
    return cls


def _encode(obj):
    if isinstance(obj, Parametrizable):
        return {"py/object": getattr(type(obj), "_ref_qualname", f"{type(obj).__module__}.{type(obj).__name__}"),
                "py/state": {k: _encode(v) for k, v in obj.data.items()}}
    if isinstance(obj, (list, tuple)):
        return [_encode(v) for v in obj]
    if isinstance(obj, (np.integer,)):
        return int(obj)
    if isinstance(obj, (np.floating,)):
        return float(obj)
    return obj


def _decode(node):
    if isinstance(node, dict) and "py/object" in node:
        cls = _REGISTRY.get(node["py/object"])
        if cls is None:
            raise ValueError(f"unknown class {node['py/object']!r} in serialized adjustment")
        obj = cls.__new__(cls)
        obj.__setstate__({k: _decode(v) for k, v in node.get("py/state", {}).items()})
        return obj
    if isinstance(node, list):
        return [_decode(v) for v in node]
    return node


class ParametrizableWithDataset(Parametrizable):
    """Parametrizable that also stores a dataset in ``ds`` (base.py:75-100)."""

    _attribute = "_xsdba_parameters"

    @classmethod
    def from_dataset(cls, ds):
        ds = as_dataset(ds)
        obj = _decode(json.loads(ds.attrs[cls._attribute]))
        obj.set_dataset(ds)
        return obj

    def set_dataset(self, ds) -> None:
        self.ds = ds
        self.ds.attrs[self._attribute] = json.dumps(_encode(self))


class Grouper(Parametrizable):
    """Grouping information (base.py:118-501): ``Grouper("time.dayofyear", window=31)``."""

    _repr_hide_params = ["dim", "prop"]
    PROP = "<PROP>"
    DIM = "<DIM>"
    ADD_DIMS = "<ADD_DIMS>"

    def __init__(self, group: str, window: int = 1, add_dims=None):
        if group == "time" and window > 1:
            raise ValueError(
                "The group given is 'time', but the window given is greater than 1. The `group = 'time'` option "
                "takes the complete series, thus the concept of window is not applicable in this case. When using "
                "`group = 'time'`, `window=1` is expected."
            )
        dim, _, prop = group.partition(".")
        if group == "5D":
            dim = "time"
        prop = prop or "group"
        add_dims = [add_dims] if isinstance(add_dims, str) else add_dims
        super().__init__(dim=dim, add_dims=list(add_dims or []), prop=prop, name=group, window=window)

    @classmethod
    def from_kwargs(cls, **kwargs):
        kwargs["group"] = cls(group=kwargs.pop("group"), window=kwargs.pop("window", 1), add_dims=kwargs.pop("add_dims", []))
        return kwargs

    @property
    def freq(self):
        return {"group": "YS", "season": "QS-DEC", "month": "MS", "week": "W", "dayofyear": "D"}.get(self.prop, None)

    @property
    def prop_name(self):
        return "year" if self.prop == "group" else self.prop

    # ---- coordinates / indexes ------------------------------------------------------------------------
    def _time_index(self, da) -> TimeIndex:
        da = as_dataarray(da) if not isinstance(da, Dataset) else da
        crd = da.coords.get(self.dim) if hasattr(da, "coords") else None
        if crd is None or crd.index is None:
            raise ValueError(f"`{self.dim}` must be a time-like coordinate of the input to group on {self.name!r}")
        return crd.index

    def get_coordinate(self, ds=None) -> DataArray:
        """Coordinate of the grouped dimension as in the output of ``group.apply`` (base.py:207-230)."""
        if self.prop == "month":
            return DataArray(np.arange(1, 13), dims=("month",), name="month")
        if self.prop == "season":
            return DataArray(np.array(["DJF", "MAM", "JJA", "SON"]), dims=("season",), name="season")
        if self.prop == "dayofyear":
            mdoy = MAX_DOY[self._time_index(ds).calendar] if ds is not None else 365
            return DataArray(np.arange(1, mdoy + 1), dims=("dayofyear",), name="dayofyear")
        if self.prop == "group":
            return DataArray(np.array([1]), dims=("group",), name="group")
        raise NotImplementedError("No grouping found.")

    def get_index(self, da, interp: bool | None = None) -> DataArray:
        """Group index of each element along the main dimension (base.py:274-345)."""
        da = as_dataarray(da) if not isinstance(da, Dataset) else da
        if self.prop == "group":
            n = da.sizes[self.dim]
            if self.dim == "time":
                return DataArray(np.ones(n, dtype=int), dims=(self.dim,), coords={self.dim: da.coords[self.dim]}, name="group")
            return da[self.dim].rename("group")
        ind = self._time_index(da)
        if interp and self.dim == "time":
            # fractional positions between group centres (what the interpolation of the factors is evaluated at)
            if self.prop == "dayofyear":
                i = ind.dayofyear
            elif self.prop == "month":
                i = ind.month - 0.5 + ind.day / ind.days_in_month
            elif self.prop == "season":
                if ind.calendar == "360_day":
                    year_len = 360
                elif ind.calendar in ("noleap", "365_day"):
                    year_len = 365
                else:
                    year_len = 365 + ind.is_leap_year
                i = ind.dayofyear / year_len * 4 - 1 / 6
            else:
                raise ValueError(f"Interpolation is not supported for {self.dim}.{self.prop}.")
        else:
            if self.prop == "season":
                i = ind.month % 12 // 3
            elif self.prop in ("month", "dayofyear", "year", "day", "week"):
                i = getattr(ind, self.prop)
            else:
                raise ValueError(f"Index {self.name} is not supported by {self.__class__.__name__}.")
            kind = np.asarray(i).dtype
            if not np.issubdtype(kind, np.integer):
                raise ValueError(f"Index {self.name} is not of type int (rather {kind}), but {self.__class__.__name__} requires integer indexes.")
        return DataArray(np.asarray(i), dims=(self.dim,), coords={self.dim: da.coords[self.dim]}, name=self.prop)

    def n_groups(self, da=None) -> int:
        return {"group": 1, "month": 12, "season": 4, "week": 53}.get(self.prop) or int(self.get_coordinate(da).size)

    def group_index(self, da):
        """(0-based int32 group of every time step, number of groups): what the HIP kernels take as
        ``group_of_time`` / ``n_groups`` (include/xsdba_hip.h)."""
        G = self.n_groups(da)
        if self.prop == "group":
            return np.zeros(as_dataarray(da).sizes[self.dim] if not isinstance(da, Dataset) else da.sizes[self.dim], np.int32), 1
        idx = np.asarray(self.get_index(da, interp=False).values)
        g0 = idx if self.prop == "season" else idx - 1
        return g0.astype(np.int32), G

    # ---- generic host application (third-party functions) ---------------------------------------------
    def apply(self, func, da, main_only: bool = False, **kwargs):
        """Apply ``func(group, dim=dims, **kwargs)`` group-wise (base.py:347-457).

        Host loop over the groups on numpy data, for user functions and string reductions; the functions of
        the accelerated path do not come through here (they are fused on the device).
        """
        is_ds = isinstance(da, (dict, Dataset)) or hasattr(da, "data_vars")
        ds = as_dataset(da) if hasattr(da, "data_vars") else (Dataset(da) if isinstance(da, dict) else None)
        one = None if is_ds else as_dataarray(da)
        ref = one if one is not None else next(v for v in ds.data_vars.values() if self.dim in v.dims)
        gidx, G = self.group_index(ref)
        T = ref.sizes[self.dim]
        W = 1 if main_only else self.window
        if main_only:
            dims = self.dim
        else:
            dims = [self.dim] + [d for d in self.add_dims if d in ref.dims]
            if self.window > 1:
                dims += ["window"]

        def windows(v: DataArray) -> DataArray:
            if W == 1 or self.dim not in v.dims:
                return v
            ax = v.dims.index(self.dim)
            x = np.moveaxis(v.values, ax, -1)
            out = np.full(x.shape + (W,), np.nan, dtype=np.result_type(x.dtype, np.float32))
            for j in range(W):  # rolling(center=True).construct: slot j holds x[t + j - W//2] (base.py:261-262)
                o = j - W // 2
                lo, hi = max(0, -o), min(T, T - o)
                out[..., lo:hi, j] = x[..., lo + o:hi + o]
            out = np.moveaxis(out, -2, ax)
            return DataArray(out, v.dims + ("window",), v.coords, v.attrs, v.name)

        # the rolling-construct array (W copies of the data) is built ONCE, the groups index into it (it used to be rebuilt for
        # every group: 365 full copies for a day-of-year grouping)
        if one is not None:
            one_w = windows(one)
        else:
            ds_w = {k: (windows(v) if self.dim in v.dims else v) for k, v in ds.data_vars.items()}
        pieces, sels = [], []
        for g in range(G):
            sel = np.nonzero(gidx == g)[0]
            if sel.size == 0:
                continue
            if one is not None:
                blk = one_w.isel({self.dim: sel})
            else:
                blk = Dataset({k: (v.isel({self.dim: sel}) if self.dim in v.dims else v) for k, v in ds_w.items()})
            if isinstance(func, str):
                res = getattr(blk, func)(dim=dims, **kwargs)
            else:
                res = func(blk, dim=dims, **kwargs)
            pieces.append((g, res))
            sels.append(sel)

        def combine(parts):
            first = parts[0]
            if self.dim in first.dims:  # not reduced: put back on the time axis, window centre only
                ax = first.dims.index(self.dim)
                parts = [p.isel(window=self.window // 2) if "window" in p.dims else p for p in parts]
                shape = list(parts[0].shape)
                shape[ax] = T
                out = np.full(shape, np.nan, dtype=np.result_type(parts[0].dtype, np.float32))
                for p, sel in zip(parts, sels):
                    idx = [slice(None)] * out.ndim
                    idx[ax] = sel
                    out[tuple(idx)] = p.values
                coords = OrderedDict(parts[0].coords)
                coords[self.dim] = ref.coords[self.dim]
                return DataArray(out, parts[0].dims, coords, parts[0].attrs, parts[0].name)
            if self.prop == "week":
                # no fixed coordinate (the reference's get_coordinate knows none either): like its groupby, the weeks present
                present = np.array([g for g, _ in pieces])
                full = np.stack([p.values for p in parts]).astype(np.result_type(first.dtype, np.float32), copy=False)
                coords = OrderedDict(first.coords)
                coords["week"] = Coordinate("week", present + 1)
                return DataArray(full, ("week",) + first.dims, coords, first.attrs, first.name)
            crd = self.get_coordinate(ref)
            full = np.full((crd.size,) + first.shape, np.nan, dtype=np.result_type(first.dtype, np.float32))
            for (g, _), p in zip(pieces, parts):
                full[g] = p.values
            coords = OrderedDict(first.coords)
            coords[self.prop] = Coordinate(self.prop, crd.values)
            return DataArray(full, (self.prop,) + first.dims, coords, first.attrs, first.name)

        if isinstance(pieces[0][1], Dataset):
            out = Dataset({k: combine([p[k] for _, p in pieces]) for k in pieces[0][1].data_vars})
        else:
            out = combine([as_dataarray(p) for _, p in pieces])
        out.attrs["group"] = self.name
        out.attrs["group_compute_dims"] = dims
        out.attrs["group_window"] = self.window
        return out

    @staticmethod
    def filter_dim(da, dim):
        """The main dimension plus those of the additional ones `da` actually has."""
        if isinstance(dim, str):
            return dim
        main, *others = dim
        return [main, *(d for d in others if d in da.dims)]

    @staticmethod
    def filter_add_dims(dim):
        """What is left of a dimension list without the additional dimensions ("window" is not one)."""
        if isinstance(dim, str):
            return [dim]
        dropped = {d for d in dim[1:] if d != "window"}
        return [d for d in dict.fromkeys(dim) if d not in dropped]


_register(Grouper, "xsdba.base")


def parse_group(func, kwargs=None, allow_only=None):
    """Make the ``group`` argument of `func` a :class:`Grouper` (base.py:504-538): a string (with the ``window`` / ``add_dims``
    given beside it) is turned into one, the default of the signature is used when none is given.  With `kwargs`: parse that
    dictionary and return it; without: return `func` wrapped so that every call is parsed."""
    par = signature(func).parameters.get("group")
    fallback = par.default if par is not None else None

    def normalise(kw):
        if fallback or "group" in kw:
            if not isinstance(kw.setdefault("group", fallback), Grouper):
                kw = Grouper.from_kwargs(**kw)
        grp = kw.get("group")
        if allow_only is not None and grp is not None and grp.prop not in allow_only:
            raise ValueError(f"Grouping on {grp.prop_name} is not allowed for this function. Should be one of {allow_only}.")
        return kw

    if kwargs is not None:
        return normalise(kwargs)

    def _parse_group(*f_args, **f_kwargs):
        return func(*f_args, **normalise(f_kwargs))

    _parse_group.__name__ = getattr(func, "__name__", "parse_group")
    _parse_group.__doc__ = func.__doc__
    _parse_group.__wrapped__ = func
    return _parse_group


def _merge_dimensions(*seqs):
    """One list holding every name of the given lists, the relative order inside each list kept: a name not seen yet goes
    right behind the last name of its list that was (base.py:585-599)."""
    merged = list(seqs[0])
    for names in seqs[1:]:
        anchor = 0  # position in `merged` of the latest name of `names` found there
        for name in names:
            if name not in merged:
                merged.insert(anchor + 1, name)
                continue
            pos = merged.index(name)
            if pos < anchor:
                raise ValueError("Dimensions order mismatch, lists are not mergeable.")
            anchor = pos
    return merged


# ---- blocks of cells (the role dask chunks play for the reference, base.py:640-702) -------------------------------------
def block_budget_bytes() -> float:
    """Largest host-backed input (all variables of one block call together) handed to the GPU in one piece; beyond it
    ``map_blocks`` cuts the non-reduced dimensions into blocks.  ``XSDBA_AMD_BLOCK_MAX_MB`` (default 65536: a quarter of an
    MI355X's HBM, which leaves room for the trained tables and the library's scratch)."""
    import os

    return float(os.environ.get("XSDBA_AMD_BLOCK_MAX_MB", "65536")) * 2 ** 20


def _nbytes(v) -> int:
    return int(np.prod(v.shape, dtype=np.int64)) * np.dtype(v.dtype).itemsize  # (never `.values.nbytes`: that would gather a sharded array)


def _plan_blocks(ds, base_dims, keep_whole=(), n_workers=1):
    """(dim, bounds, contexts, keep) when the variables of `ds` are to be cut along a dimension the block function neither
    reduces nor creates — never the grouping dimension (`keep_whole`: ranks, trends and windows need the whole series) —, else
    None (device-resident inputs are already where they are used):
      * a variable that is ALREADY sharded over the devices (a table trained by a multi-device call: labeled.ShardedDataArray)
        dictates the cut — its parts' bounds — and where every block runs: on the context that holds its part;
      * host-backed variables beyond the block budget are cut to fit it (results are brought back to the host: keep = False);
      * with several devices to share them (`n_workers` > 1 and at least ``XSDBA_AMD_MULTI_MIN_MB``, default 64, of series) into
        two blocks per device, so that a device's next upload runs under another one's kernels; the results stay on the device
        that computed them (keep = True) and come back as sharded arrays."""
    import os

    from .labeled import ShardedDataArray

    base_dims = [d for d in base_dims if d not in keep_whole]
    variables = list(ds.data_vars.values()) if isinstance(ds, Dataset) else [ds]
    if any(getattr(v, "device_data", None) is not None for v in variables):
        return None
    sharded = [v for v in variables if isinstance(v, ShardedDataArray) and v.part_dim in base_dims and getattr(v, "_host", None) is None]
    if sharded:
        lead = sharded[0]
        bounds = [(a, b_) for a, b_, _ in lead.parts]
        ctxs = [getattr(getattr(p, "device_data", None), "ctx", None) for _, _, p in lead.parts]
        same = all(v.part_dim == lead.part_dim and [(a, b_) for a, b_, _ in v.parts] == bounds for v in sharded)
        sizes_ok = all(v.sizes[lead.part_dim] == lead.sizes[lead.part_dim] for v in variables if lead.part_dim in v.dims)
        if same and sizes_ok and all(c is not None for c in ctxs):
            return lead.part_dim, bounds, ctxs, True
        # (parts that do not line up, or no longer on a device: gathered on the host and planned afresh below)
    cut = [v for v in variables if any(d in base_dims and v.sizes[d] > 1 for d in v.dims)]
    total = sum(_nbytes(v) for v in cut)
    budget = block_budget_bytes()
    share = n_workers > 1 and total >= float(os.environ.get("XSDBA_AMD_MULTI_MIN_MB", "64")) * 2 ** 20
    if not cut or (total <= budget and not share):
        return None
    big = max(cut, key=_nbytes)
    cands = [d for d in big.dims if d in base_dims and big.sizes[d] > 1]
    # the outermost such dimension: its blocks are contiguous pieces of a C-ordered cube
    dim = cands[0]
    n = big.sizes[dim]
    step = max(1, int(n * budget // total)) if total > budget else n
    if share:
        step = min(step, -(-n // (2 * n_workers)))
    if step >= n:
        return None
    bounds = [(i0, min(n, i0 + step)) for i0 in range(0, n, step)]
    return dim, bounds, None, share and total <= budget


_block_pools: dict = {}


def _block_workers():
    """(contexts, executors): one single-thread executor per worker context — a context is only ever driven by its own thread
    (the C ABI: contexts are independent, one context is not re-entrant), ctypes releases the GIL for the length of a call."""
    from concurrent.futures import ThreadPoolExecutor

    from ._lib import worker_contexts

    ctxs = worker_contexts()
    key = tuple(id(c) for c in ctxs)
    if key not in _block_pools:
        _block_pools[key] = [ThreadPoolExecutor(max_workers=1, thread_name_prefix=f"xsdba-amd-w{i}-dev{c.device}") for i, c in enumerate(ctxs)]
    return ctxs, _block_pools[key]


def _isel_block(ds, dim, sl):
    def cut(v):
        return v.isel({dim: sl}) if dim in v.dims else v

    if isinstance(ds, Dataset):
        return Dataset({k: cut(v) for k, v in ds.data_vars.items()}, attrs=ds.attrs)
    return cut(ds)


def _concat_blocks(outs, dim):
    """the blocks' results as one Dataset: host arrays are concatenated; results that stayed on their devices become
    labeled.ShardedDataArray (nothing moves until somebody asks for `values`)"""
    from .labeled import ShardedDataArray

    first = outs[0]
    merged = {}
    for name, v0 in first.data_vars.items():
        if dim not in v0.dims:
            merged[name] = v0
            continue
        parts = [o.data_vars[name] for o in outs]
        if any(getattr(p, "device_data", None) is not None and getattr(p, "_host", None) is None for p in parts):
            merged[name] = ShardedDataArray(parts, dim, v0.attrs, name)
            continue
        ax = v0.dims.index(dim)
        host = [np.asarray(p.transpose(*v0.dims).values) for p in parts]
        coords = OrderedDict(v0.coords)
        if dim in v0.coords:
            coords[dim] = Coordinate(dim, np.concatenate([np.asarray(p.coords[dim].values) for p in parts]), v0.coords[dim].attrs)
        merged[name] = DataArray(np.concatenate(host, axis=ax), v0.dims, coords, v0.attrs, name)
    return Dataset(merged, attrs=first.attrs)


def map_blocks(reduces=None, **out_vars):
    """Declare a block function ``func(ds, **kwargs) -> Dataset`` (base.py:563-726).

    Data here are numpy-backed (one block), so the wrapper's job is the contract around the call: group
    parsing, the placeholder dims ``<PROP>/<DIM>/<ADD_DIMS>``, the output dimension order
    (untouched dims first, new dims last, :653-656), the output dtype (largest input dtype, :681-685),
    dropping auxiliary coordinates (:708-712) and re-raising any failure as
    ``ValueError("<func> failed on block with coords : ...")`` (:698-703).  ``.func`` exposes the raw body.
    """
    out_dims = _merge_dimensions(*out_vars.values()) if out_vars else []
    red_dims = list(reduces or [])

    def _decorator(func):
        def _map_blocks(ds, **kwargs):
            ds = as_dataset(ds) if (isinstance(ds, Dataset) or hasattr(ds, "data_vars")) else as_dataarray(ds)
            group = kwargs.get("group")
            if {Grouper.PROP, Grouper.DIM, Grouper.ADD_DIMS}.intersection(out_dims + red_dims) and group is None:
                raise ValueError("Missing required `group` argument.")
            placeholders = {Grouper.PROP: [group.prop], Grouper.DIM: [group.dim], Grouper.ADD_DIMS: group.add_dims} if group is not None else {}
            all_in = set(ds.sizes) if isinstance(ds, Dataset) else set(ds.dims)
            # (an adjust call may see none of the pooled dims: sim need not have them, and unlike the reference's the trained
            # tables here never carry them — its P0_ref / P0_hist / pth dummies do, base.py:52-58)
            adjusting = isinstance(ds, Dataset) and "sim" in ds and "af" in ds
            if group is not None and group.add_dims and not adjusting and not set(group.add_dims).issubset(all_in):
                raise ValueError("`add_dims` argument needs to be a dimension in one of the input datasets.")
            new_dims, reduced_dims = [], []
            for dim in out_dims:
                new_dims.extend(placeholders.get(dim, [dim]))
            for dim in red_dims:
                reduced_dims.extend(placeholders.get(dim, [dim]))
            in_dims = list(ds.dims) if not isinstance(ds, Dataset) else list(ds.sizes)
            base_dims = [d for d in in_dims if d not in new_dims and d not in reduced_dims]
            all_dims = base_dims + new_dims
            for dim in new_dims:
                if group is not None and dim in (group.prop, group.dim):
                    continue
                if dim not in kwargs and dim not in in_dims:
                    raise ValueError(f"This function adds the {dim} dimension, its coordinate must be provided as a keyword argument.")
            try:
                from ._lib import use_context, worker_devices

                n_workers = len(worker_devices())
                plan = _plan_blocks(ds, base_dims, keep_whole=(group.dim,) if group is not None else ("time",), n_workers=n_workers)
                if plan is None:
                    out = as_dataset(func(ds, **kwargs))
                else:
                    # inputs larger than the block budget, or several devices to share them: the non-reduced dimension is cut in
                    # blocks, every block is one call (uploaded through page-locked pieces, computed), like dask chunks — dealt
                    # round-robin over one context per device (XSDBA_AMD_DEVICES), each driven by its own thread; a block whose
                    # tables already sit on a device runs there
                    import threading

                    from ._lib import _tls
                    from .labeled import to_device_if_large

                    bdim, bounds, part_ctxs, keep = plan
                    seeds = {"lock": threading.Lock(), "base": None}  # (processing._draw_seed: one base for all blocks)

                    def one_block(b, ctx=None):
                        def body():
                            blk = _isel_block(ds, bdim, slice(*bounds[b]))
                            blk = (Dataset({k: to_device_if_large(v) for k, v in blk.data_vars.items()}, attrs=blk.attrs)
                                   if isinstance(blk, Dataset) else to_device_if_large(blk))
                            o = as_dataset(func(blk, **kwargs))
                            if keep:
                                return o  # (the block's results stay on its device: _concat_blocks makes sharded arrays of them)
                            return Dataset({k: DataArray(np.asarray(v.values), v.dims, v.coords, v.attrs, v.name)
                                            for k, v in o.data_vars.items()}, attrs=o.attrs)

                        if ctx is None:
                            return body()
                        with use_context(ctx):
                            _tls.seeds = {"shared": seeds, "block": b, "k": 0}
                            try:
                                return body()
                            finally:
                                _tls.seeds = None

                    if n_workers > 1 or part_ctxs is not None:
                        ctxs, pools = _block_workers()
                        by_ctx = {id(c): i for i, c in enumerate(ctxs)}
                        if part_ctxs is not None and not all(id(c) in by_ctx for c in part_ctxs):
                            # tables sharded over contexts that are no longer the workers': run every block where its part lives,
                            # one after the other
                            outs = [one_block(b, part_ctxs[b]) for b in range(len(bounds))]
                        else:
                            slot = [by_ctx[id(part_ctxs[b])] if part_ctxs is not None else b % len(pools) for b in range(len(bounds))]
                            futs = [pools[slot[b]].submit(one_block, b, ctxs[slot[b]]) for b in range(len(bounds))]
                            outs = [f.result() for f in futs]
                    else:
                        outs = [one_block(b) for b in range(len(bounds))]
                    out = _concat_blocks(outs, bdim)
                def _ordered(v):
                    # the template order of the reference (base.py:696-700), for device-resident results too: their transpose is a
                    # relabelling (labeled.DataArray.transpose), the buffer stays as the kernels wrote it
                    # (a variable may keep dimensions the template reduces: P0_ref / P0_hist / pth keep the `add_dims`, _adjustment.py:86-95
                    #  of the reference — [PROP, ADD_DIMS]: they follow the template's)
                    return v.transpose(*[d for d in all_dims if d in v.dims], *[d for d in v.dims if d not in all_dims])

                out = Dataset({k: _ordered(v) for k, v in out.data_vars.items()}, attrs=out.attrs)
            except Exception as err:
                crds = list(ds.coords)
                raise ValueError(f"{func.__name__} failed on block with coords : {crds}.") from err
            return out

        wrapped = parse_group(_map_blocks)
        wrapped.__name__ = getattr(func, "__name__", "map_blocks")
        wrapped.__doc__ = func.__doc__
        wrapped.__dict__["func"] = func
        return wrapped

    return _decorator


def map_groups(reduces=None, main_only: bool = False, fused=None, **out_vars):
    """Declare a per-group function ``func(ds, dim, **kwargs) -> Dataset`` (base.py:729-778).

    ``fused`` (extension of this implementation): a callable ``fused(ds, group=..., **kwargs) -> Dataset``
    that produces the result for ALL groups at once — the functions of the accelerated path pass their HIP
    implementation here; without it the wrapper falls back on ``group.apply(func, ...)`` like the reference.
    """
    if not reduces:
        reduces = [Grouper.DIM] if main_only else [Grouper.DIM, Grouper.ADD_DIMS]

    def _run_all_groups(func, dsblock, group, kwargs):
        """the block's result for every group: the device implementation when the function has one, else the host loop"""
        if fused is None:
            return group.apply(func, dsblock, main_only=main_only, **kwargs)
        out = fused(dsblock, group=group, **kwargs)
        out.attrs.update(group=group.name, group_window=group.window,
                         group_compute_dims=[group.dim, "window"] if group.window > 1 else [group.dim])
        return out

    def _decorator(func):
        name = getattr(func, "__name__", "func")

        def per_block(dsblock, group, **kwargs):
            return _run_all_groups(func, dsblock, group, kwargs)

        per_block.__name__ = f"group_{name}"  # (the name the failure message of map_blocks reports, as the reference's does)
        mapped = map_blocks(reduces=reduces, **out_vars)(per_block)
        mapped.__name__, mapped.__doc__ = name, func.__doc__
        mapped.__dict__["func"] = func  # the raw per-group body stays reachable (base.py:776)
        return mapped

    return _decorator


def uses_dask(*das) -> bool:
    return False
