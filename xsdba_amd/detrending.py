"""Detrending objects of the hot path (xsdba.detrending, detrending.py:17-131 BaseDetrend, :165-208 PolyDetrend).

``PolyDetrend(degree 0..6, preserve_mean)`` is computed by the HIP group-trend kernel (NaN-skipping centred window means
per time step, per-group least squares against the float64 time axis); other detrenders of the reference
(Loess, RollingMean, Mean, No) are outside the accelerated path."""
from __future__ import annotations

import warnings

import numpy as np

from . import hip
from .base import Grouper, ParametrizableWithDataset, _register, parse_group
from .labeled import DataArray, as_dataarray
from .utils import ADDITIVE, cell_time_view


class BaseDetrend(ParametrizableWithDataset):
    """fit(da) -> fitted copy with ``ds.trend``; detrend(da); retrend(da) (detrending.py:17-131)."""

    def __init__(self, *, group="time", kind="+", mult_skip_zeros=False, **kwargs):
        kw = parse_group(self.__class__.__init__, dict(group=group))
        super().__init__(group=kw["group"], kind=kind, mult_skip_zeros=mult_skip_zeros, **kwargs)

    @property
    def fitted(self):
        return hasattr(self, "ds")

    def fit(self, da):
        da = as_dataarray(da)
        new = self.__class__(**self.parameters)
        trend = new._get_trend(da).astype(da.dtype).rename("trend")
        new.set_dataset(trend.to_dataset())
        if "units" in da.attrs:
            new.ds.trend.attrs["units"] = da.attrs["units"]
        return new

    def detrend(self, da):
        if not self.fitted:
            raise ValueError("You must call fit() before detrending.")
        return self._detrend(as_dataarray(da), self.ds.trend)

    def retrend(self, da):
        if not self.fitted:
            raise ValueError("You must call fit() before retrending")
        return self._retrend(as_dataarray(da), self.ds.trend)

    def _apply(self, da, trend, inverted, skip_zeros):
        """da (+|*) trend, or invert(trend), on the device: both as [cell, time] blocks in the dimension order of `da`"""
        dim = self.group.dim
        trend = trend.transpose(*[d for d in da.dims if d in trend.dims])
        x2, cell_dims, cell_shape = cell_time_view(da, dim)
        dt = x2.dtype if x2.dtype.kind == "f" else np.float64
        if set(trend.dims) == {dim}:
            f2 = trend.values
        else:
            missing = [d for d in cell_dims if d not in trend.dims]
            if missing:
                raise ValueError(f"trend lacks dimension(s) {missing} of the data")
            f2, _, _ = cell_time_view(trend.transpose(*cell_dims, dim), dim)
            if type(f2).__name__ == "DeviceArray":  # (the factor of series_correction is a host array, [cell, time])
                h = f2.download().reshape(f2.shape)
                f2 = h.T if getattr(f2, "cell_time_strides", None) is not None else h
        out = hip.series_correction(x2.astype(dt, copy=False), f2.astype(dt, copy=False), self.kind, inverted, skip_zeros)
        coords = {k: c for k, c in da.coords.items() if k in cell_dims or k == dim}
        return DataArray(out.reshape(cell_shape + (out.shape[-1],)), cell_dims + (dim,), coords, dict(da.attrs), da.name).transpose(*da.dims)

    def _detrend(self, da, trend):
        if self.mult_skip_zeros and self.kind != "*":
            warnings.warn(f"mult_skip_zeros is only used for kind='*'. Your kind is {self.kind}. mult_skip_zeros will be ignored.",
                          UserWarning, stacklevel=2)
        return self._apply(da, trend, True, bool(self.mult_skip_zeros) and self.kind == "*")

    def _retrend(self, da, trend):
        return self._apply(da, trend, False, False)

    def _get_trend(self, da):
        raise NotImplementedError

    def __repr__(self):
        rep = super().__repr__()
        return rep if self.fitted else f"<{rep} | unfitted>"


class PolyDetrend(BaseDetrend):
    """Detrend time series using a polynomial regression per group (detrending.py:165-208)."""

    def __init__(self, group="time", kind=ADDITIVE, degree=4, preserve_mean=False, mult_skip_zeros=False):
        super().__init__(group=group, kind=kind, degree=degree, preserve_mean=preserve_mean, mult_skip_zeros=mult_skip_zeros)

    def _get_trend(self, da: DataArray) -> DataArray:
        group: Grouper = self.group
        if not 0 <= int(self.degree) <= 6:
            raise NotImplementedError(f"PolyDetrend(degree={self.degree}) is not implemented on the HIP path (0..6)")
        if group.add_dims:
            raise NotImplementedError("PolyDetrend with `add_dims` is only available through dqm_adjust on the HIP path")
        x2, cell_dims, cell_shape = cell_time_view(da, group.dim)
        gidx, G = group.group_index(da)
        tnum = group._time_index(da).numeric_ns
        tr = hip.polydetrend_trend(x2, gidx, G, group.window, tnum, int(self.degree), bool(self.preserve_mean), self.kind)
        coords = {k: c for k, c in da.coords.items() if k in cell_dims or k == group.dim}
        return DataArray(tr.reshape(cell_shape + (tr.shape[-1],)), cell_dims + (group.dim,), coords)


class NoDetrend(BaseDetrend):
    """Convenience class for polymorphism; does nothing (detrending.py:134-147)."""

    def _get_trend(self, da: DataArray) -> DataArray:
        return da.isel({self.group.dim: 0})

    def _detrend(self, da, trend):
        return da

    def _retrend(self, da, trend):
        return da


class MeanDetrend(BaseDetrend):
    """Removes only the (group) mean from the data (detrending.py:150-162): the NaN-skipping mean of every (cell, group)
    block over time and window (``k_group_mean``), broadcast on the group of each time step (``k_scale_groups``)."""

    def _get_trend(self, da: DataArray) -> DataArray:
        group: Grouper = self.group
        if group.add_dims:
            raise NotImplementedError("MeanDetrend with `add_dims` is not implemented on the HIP path")
        x2, cell_dims, cell_shape = cell_time_view(da, group.dim)
        x2 = x2.astype(np.result_type(x2.dtype, np.float32), copy=False)
        gidx, G = group.group_index(da)
        means = hip.group_means(x2, gidx, G, group.window)
        tr = hip.scaling_adjust(np.zeros_like(x2), means, gidx if G > 1 else None, "+")
        coords = {k: c for k, c in da.coords.items() if k in cell_dims or k == group.dim}
        return DataArray(tr.reshape(cell_shape + (tr.shape[-1],)), cell_dims + (group.dim,), coords)


class RollingMeanDetrend(BaseDetrend):
    """Detrend time series using a rolling mean (detrending.py:298-356), on the whole series (``group="time"``: the grouped
    variants roll over the steps of each group, which is not on the HIP path; neither are ``weights``)."""

    def __init__(self, group="time", kind=ADDITIVE, win=30, weights=None, min_periods=None, mult_skip_zeros=False):
        if weights is not None:
            raise NotImplementedError("RollingMeanDetrend(weights=...) is not implemented on the HIP path")
        super().__init__(group=group, kind=kind, win=win, weights=weights, min_periods=min_periods, mult_skip_zeros=mult_skip_zeros)

    def _get_trend(self, da: DataArray) -> DataArray:
        group: Grouper = self.group
        if group.prop != "group" or group.add_dims:
            raise NotImplementedError("RollingMeanDetrend is implemented for group='time' on the HIP path")
        x2, cell_dims, cell_shape = cell_time_view(da, group.dim)
        x2 = x2.astype(np.result_type(x2.dtype, np.float32), copy=False)
        tr = hip.rolling_mean(x2, int(self.win), self.min_periods)
        coords = {k: c for k, c in da.coords.items() if k in cell_dims or k == group.dim}
        return DataArray(tr.reshape(cell_shape + (tr.shape[-1],)), cell_dims + (group.dim,), coords)


_register(PolyDetrend, "xsdba.detrending")
_register(RollingMeanDetrend, "xsdba.detrending")
_register(NoDetrend, "xsdba.detrending")
_register(MeanDetrend, "xsdba.detrending")
