"""Quantile kernels with the reference's names (xsdba.nbutils, nbutils.py:113-148, 224-271): the numba JIT
loops are replaced by the HIP sort + type-7 nanquantile kernel."""
from __future__ import annotations

import numpy as np

from . import hip
from .labeled import Coordinate, DataArray, as_dataarray


def quantile(da, q, dim):
    """Quantiles from a fixed list `q` along `dim` (one or several dimensions), NaN-skipping, type 7.

    Output keeps the other dimensions and gains a trailing ``quantiles`` dimension (nbutils.py:224-271).
    """
    da = as_dataarray(da)
    dims = [dim] if isinstance(dim, str) else list(dim)
    keep = [d for d in da.dims if d not in dims]
    v = da.transpose(*keep, *dims).values
    x2 = v.reshape((int(np.prod(v.shape[: len(keep)], dtype=np.int64)), -1))
    x2 = x2.astype(np.result_type(x2.dtype, np.float32), copy=False)
    qc = np.array(q, dtype=x2.dtype)  # nbutils.py:253
    out = hip.quantile_groups(x2, None, 1, 1, qc)[:, 0, :]
    coords = {k: c for k, c in da.coords.items() if k in keep}
    coords["quantiles"] = Coordinate("quantiles", np.asarray(q))
    return DataArray(out.reshape(tuple(da.sizes[d] for d in keep) + (qc.size,)), tuple(keep) + ("quantiles",), coords, da.attrs, da.name)


def _quantile(arr, q, nreduce=None):
    """Array-level form (nbutils.py:206-221): reduces the last `nreduce` axes."""
    arr = np.asarray(arr)
    nreduce = nreduce or arr.ndim
    lead = arr.shape[: arr.ndim - nreduce]
    x2 = arr.reshape((int(np.prod(lead, dtype=np.int64)) if lead else 1, -1))
    x2 = x2.astype(np.result_type(x2.dtype, np.float32), copy=False)
    out = hip.quantile_groups(x2, None, 1, 1, np.asarray(q, x2.dtype))[:, 0, :]
    return out.reshape(tuple(lead) + (len(q),)) if lead else out[0]


def vecquantiles(da, rnk, dim):
    """For when the quantile (rnk) is different for each point (nbutils.py:164-195): `da` and `rnk` share all
    dimensions but `dim`; returns an array shaped like `rnk`."""
    da, rnk = as_dataarray(da), as_dataarray(rnk)
    dims = [dim] if isinstance(dim, str) else list(dim)
    v = da.transpose(*rnk.dims, *dims).values
    x2 = v.reshape((int(np.prod(v.shape[: rnk.ndim], dtype=np.int64)), -1))
    x2 = x2.astype(np.result_type(x2.dtype, np.float32), copy=False)
    out = hip.vecquantiles(x2, rnk.values.reshape(-1).astype(x2.dtype))
    return DataArray(out.reshape(rnk.shape).astype(da.dtype), rnk.dims, rnk.coords, da.attrs)
