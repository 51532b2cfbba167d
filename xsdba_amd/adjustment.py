"""Adjustment classes of the hot path with the reference's interface (xsdba.adjustment:
BaseAdjustment :68-206, TrainAdjust :209-332, EmpiricalQuantileMapping :414-528, DetrendedQuantileMapping
:531-671, QuantileDeltaMapping :674-742).

Same call pattern — ``Cls.train(ref, hist, **kw)`` then ``obj.adjust(sim, **kw)`` —, same checks and errors,
same trained dataset (``af``, ``hist_q``, ``hist_q_raw``[, ``scaling``], attrs ``group``,
``group_compute_dims``, ``group_window``, ``_xsdba_adjustment``, ``adj_params``) and the same attributes on
``scen``.  Inputs may be xarray DataArrays (duck-typed) or ``xsdba_amd.DataArray``; when the inputs are
xarray objects the outputs are converted back to xarray.
"""
from __future__ import annotations

import datetime as dt
from warnings import warn

import numpy as np

from . import __version__
from ._adjustment import dqm_adjust, dqm_train, eqm_train, qdm_adjust, qm_adjust
from .base import Grouper, ParametrizableWithDataset, _register, parse_group
from .detrending import PolyDetrend  # noqa: F401  (re-exported like the reference)
from .labeled import DataArray, Dataset, as_dataarray, as_dataset, is_xarray
from .options import EXTRA_OUTPUT, OPTIONS
from .utils import ADDITIVE, equally_spaced_nodes

__all__ = ["BaseAdjustment", "DetrendedQuantileMapping", "EmpiricalQuantileMapping", "LOCI", "QuantileDeltaMapping", "Scaling", "TrainAdjust"]


def gen_call_string(funcname: str, *args, **kwargs) -> str:
    """Signature string for the history attribute (formatting.py:158-202)."""
    elements = []
    for name, val in [(None, a) for a in args] + list(kwargs.items()):
        if isinstance(val, DataArray):
            rep = val.name or "<array>"
        elif isinstance(val, (int, float, str, bool)) or val is None:
            rep = repr(val)
        else:
            rep = repr(val)
            if len(rep) > 50:
                rep = f"<{type(val).__name__}>"
        elements.append(f"{name}={rep}" if name is not None else rep)
    return f"{funcname}({', '.join(elements)})"


def update_history(hist_str: str, *inputs, new_name: str | None = None) -> str:
    """Timestamped history entry followed by the inputs' histories (formatting.py:69-116)."""
    merged = "\n".join((f"{i.name}: " if getattr(i, "name", None) else "") + i.attrs.get("history", "") for i in inputs
                       if i.attrs.get("history") is not None and (len(inputs) > 1 or "history" in i.attrs))
    return f"[{dt.datetime.now():%Y-%m-%d %H:%M:%S}] {new_name or ''}: {hist_str} - xsdba version: {__version__}\n" + merged


def _device_resident(da):
    """Large host series are moved to the GPU ONCE, before the adjustment code sees them (labeled.to_device_if_large) —
    unless ``map_blocks`` will cut them into blocks of cells: because they are that large (base.block_budget_bytes) or because
    several devices share the work (_lib.worker_devices): those stay on the host and every block is uploaded on its own, to
    the device that computes it."""
    import os

    from ._lib import worker_devices
    from .base import block_budget_bytes
    from .labeled import to_device_if_large

    if getattr(da, "device_data", None) is None:
        if da.values.nbytes > block_budget_bytes() / 3:
            return da
        if len(worker_devices()) > 1 and da.values.nbytes >= float(os.environ.get("XSDBA_AMD_MULTI_MIN_MB", "64")) * 2 ** 20 / 3:
            return da
    return to_device_if_large(da)


class BaseAdjustment(ParametrizableWithDataset):
    """Base class: input checks shared by every adjustment (adjustment.py:68-206)."""

    _allow_diff_calendars = True
    _allow_diff_training_times = True
    _allow_diff_time_sizes = True
    _allow_5d_grouping = False
    _attribute = "_xsdba_adjustment"

    def __init__(self, *args, _trained=False, **kwargs):
        if _trained:
            super().__init__(*args, **kwargs)
        else:
            raise ValueError("Adjustment object should be initialized through their `train` or `adjust` methods.")

    @classmethod
    def _check_inputs(cls, *inputs, group):
        if not cls._allow_5d_grouping and group.name == "5D":
            raise NotImplementedError("`group=Grouper('5D', window)` is a special grouping currently only supported for MBCn.")
        calendars = {group._time_index(inda).calendar for inda in inputs}
        if not cls._allow_diff_calendars and len(calendars) > 1:
            raise ValueError(f"Inputs are defined on different calendars, this is not supported for {cls.__name__} adjustment.")
        if group.prop == "dayofyear" and ("default" in calendars or "standard" in calendars):
            warn("Strange results could be returned when using `dayofyear` grouping on data defined in the "
                 "'proleptic_gregorian' calendar.", stacklevel=2)

    @classmethod
    def _harmonize_units(cls, *inputs, target=None):
        """Bring the inputs to common units: those of the first one, or `target` (adjustment.py:182-185, units.py:269-321);
        same units => no-op, else converted on the device (xsdba_amd/units.py)."""
        from .units import convert_units_to

        for inda in inputs:
            if "units" not in inda.attrs:
                raise ValueError(f"Input {inda.name or '<array>'!s} has no `units` attribute (required like in the reference)")
        if target is None or target == "":
            target = inputs[0].attrs["units"]
        return tuple(inda if inda.attrs["units"] == target else convert_units_to(inda, target) for inda in inputs), target

    @classmethod
    def _check_matching_times(cls, ref, hist):
        if not ref.coords["time"].equals(hist.coords["time"]):
            raise ValueError(f"`ref` and `hist` have distinct time arrays, this is not supported for {cls.__name__} adjustment.")

    @classmethod
    def _check_matching_time_sizes(cls, *inputs):
        if len({inp.sizes["time"] for inp in inputs}) > 1:
            raise ValueError(f"Inputs have different size for the time array, this is not supported for {cls.__name__} adjustment.")

    @classmethod
    def _train(cls, ref, hist, **kwargs):
        raise NotImplementedError()

    def _adjust(self, sim, *args, **kwargs):
        raise NotImplementedError()


class TrainAdjust(BaseAdjustment):
    """Adjustment objects obeying the train-adjust scheme (adjustment.py:209-332).

    Subclasses implement ``_train(ref, hist, **kwargs) -> (Dataset, params)`` and
    ``_adjust(sim, **kwargs) -> DataArray | Dataset``.
    """

    _allow_diff_calendars = True
    _attribute = "_xsdba_adjustment"
    _repr_hide_params = ["hist_calendar", "train_units"]

    @classmethod
    def _vet_training_pair(cls, ref, hist, group, checked: bool):
        """The checks of `train` (adjustment.py:240-256): calendars and grouping, one unit for both series, and — per class
        flag — identical time axes or at least equally long ones.  Returns the harmonized pair and the units they are in."""
        units = ""
        if checked:
            if group is not None:
                cls._check_inputs(ref, hist, group=group)
            (ref, hist), units = cls._harmonize_units(ref, hist)
        if not cls._allow_diff_training_times:
            cls._check_matching_times(ref, hist)
        elif not cls._allow_diff_time_sizes:
            cls._check_matching_time_sizes(ref, hist)
        return ref, hist, units

    @classmethod
    def train(cls, ref, hist, **kwargs):
        """Train the adjustment object (adjustment.py:227-269): parse the group, check the pair, run the class's `_train` and
        keep its dataset + parameters (and the calendar / units of the training data) on the new object."""
        want_xr = is_xarray(ref)
        ref, hist = (_device_resident(as_dataarray(x)) for x in (ref, hist))
        kwargs = parse_group(cls._train, kwargs)
        group = kwargs.get("group")
        ref, hist, train_units = cls._vet_training_pair(ref, hist, group, not kwargs.pop("skip_input_checks", False))
        ds, params = cls._train(ref, hist, **kwargs)
        calendar = "standard" if group is None else group._time_index(hist).calendar
        obj = cls(_trained=True, hist_calendar=calendar, train_units=train_units, **params)
        obj._xr = want_xr
        obj.set_dataset(ds)
        return obj

    def _describe_adjust_call(self, kwargs) -> str:
        return f"{self!s}.adjust(sim, {gen_call_string('', **kwargs)[1:-1]})"

    def adjust(self, sim, *args, **kwargs):
        """Return bias-adjusted data (adjustment.py:271-316): `sim` in the training units through the class's `_adjust`; the
        result carries sim's attributes, the call in `history` / `bias_adjustment`, and the training units."""
        want_xr = is_xarray(sim)
        sim = _device_resident(as_dataarray(sim))
        if not kwargs.pop("skip_input_checks", False):
            if "group" in self:
                self._check_inputs(sim, *args, group=self.group)
            (sim, *args), _ = self._harmonize_units(sim, *args, target=self.train_units)
        out = self._adjust(sim, *args, **kwargs)
        if isinstance(out, DataArray):
            out = out.rename("scen").to_dataset()
        scen = out["scen"]
        scen.attrs.update(sim.attrs)
        for name in set(sim.coords) & set(scen.coords):
            scen.coords[name].attrs.update(sim.coords[name].attrs)
        call = self._describe_adjust_call(kwargs)
        scen.attrs.update(history=update_history(f"Bias-adjusted with {call}", sim), bias_adjustment=call, units=self.train_units)
        result = out if OPTIONS[EXTRA_OUTPUT] else scen
        return result.to_xarray() if want_xr else result

    def set_dataset(self, ds):
        """Store the trained dataset in ``ds`` with the jsonpickle + repr attributes (adjustment.py:318-325)."""
        ds = as_dataset(ds)
        super().set_dataset(ds)
        self.ds.attrs["adj_params"] = str(self)

    @classmethod
    def _train(cls, ref, hist, **kwargs):
        raise NotImplementedError()

    def _adjust(self, sim, **kwargs):
        raise NotImplementedError()


def _nodes(nquantiles, dtype):
    if np.isscalar(nquantiles):
        return equally_spaced_nodes(nquantiles).astype(dtype)  # adjustment.py:480-483
    return np.asarray(nquantiles).astype(dtype)


_AF_ATTRS = dict(standard_name="Adjustment factors", long_name="Quantile mapping adjustment factors")


class EmpiricalQuantileMapping(TrainAdjust):
    """Empirical Quantile Mapping bias-adjustment (adjustment.py:414-528): F^-1_ref(F_hist(sim))."""

    _allow_diff_calendars = False
    _allow_diff_training_times = False

    @classmethod
    def _train(cls, ref, hist, *, nquantiles=20, kind: str = ADDITIVE, group="time", adapt_freq_thresh=None,
               jitter_under_thresh_value=None, jitter_over_thresh_value=None, jitter_over_thresh_upper_bnd=None,
               max_tail_factor=None):
        quantiles = _nodes(nquantiles, np.result_type(ref.dtype, np.float32))
        ds = eqm_train(Dataset({"ref": ref, "hist": hist}), group=group, kind=kind, quantiles=quantiles,
                       adapt_freq_thresh=adapt_freq_thresh, jitter_under_thresh_value=jitter_under_thresh_value,
                       jitter_over_thresh_value=jitter_over_thresh_value,
                       jitter_over_thresh_upper_bnd=jitter_over_thresh_upper_bnd, max_tail_factor=max_tail_factor)
        ds["af"].attrs.update(_AF_ATTRS)
        ds["hist_q"].attrs.update(standard_name="Model quantiles",
                                  long_name="Quantiles of model on the reference period, after preprocess")
        ds["hist_q_raw"].attrs.update(standard_name="Model quantiles",
                                      long_name="Quantiles of model on the reference period, before preprocess")
        return ds, {"group": group, "kind": kind, "adapt_freq_thresh": adapt_freq_thresh, "max_tail_factor": max_tail_factor}

    def _adjust(self, sim, interp="nearest", extrapolation="constant"):
        return qm_adjust(self.ds.assign(sim=sim), group=self.group, interp=interp, extrapolation=extrapolation,
                         kind=self.kind, adapt_freq_thresh=self.adapt_freq_thresh, max_tail_factor=self.max_tail_factor)["scen"]


class DetrendedQuantileMapping(TrainAdjust):
    """Detrended Quantile Mapping bias-adjustment (adjustment.py:531-671; Cannon et al. 2015)."""

    _allow_diff_calendars = False
    _allow_diff_training_times = False

    @classmethod
    def _train(cls, ref, hist, *, nquantiles=20, kind: str = ADDITIVE, group="time", adapt_freq_thresh=None,
               jitter_under_thresh_value=None, jitter_over_thresh_value=None, jitter_over_thresh_upper_bnd=None,
               max_tail_factor=None):
        if group.prop not in ["group", "dayofyear"]:
            warn(f"Using DQM with a grouping other than 'dayofyear' is not recommended (received {group.name}).", stacklevel=2)
        quantiles = _nodes(nquantiles, np.result_type(ref.dtype, np.float32))
        ds = dqm_train(Dataset({"ref": ref, "hist": hist}), group=group, quantiles=quantiles, kind=kind,
                       adapt_freq_thresh=adapt_freq_thresh, jitter_under_thresh_value=jitter_under_thresh_value,
                       jitter_over_thresh_value=jitter_over_thresh_value,
                       jitter_over_thresh_upper_bnd=jitter_over_thresh_upper_bnd, max_tail_factor=max_tail_factor)
        ds["af"].attrs.update(_AF_ATTRS)
        ds["hist_q"].attrs.update(
            standard_name="Model quantiles",
            long_name="Quantiles of the anomalies of the model on the reference period, after preprocessing steps. ")
        ds["hist_q_raw"].attrs.update(standard_name="Model quantiles",
                                      long_name="Quantiles of model on the reference period, befofe the preprocessing steps.")
        ds["scaling"].attrs.update(standard_name="Scaling factor",
                                   description="Scaling factor making the mean of hist match the one of hist.")
        return ds, {"group": group, "kind": kind, "adapt_freq_thresh": adapt_freq_thresh, "max_tail_factor": max_tail_factor}

    def _adjust(self, sim, interp="nearest", extrapolation="constant", detrend=1):
        # (.scen is all that is kept, like the reference: the block function is told so and need not write the trend array)
        out = dqm_adjust(self.ds.assign(sim=sim), interp=interp, extrapolation=extrapolation, detrend=detrend,
                         group=self.group, kind=self.kind, adapt_freq_thresh=self.adapt_freq_thresh,
                         max_tail_factor=self.max_tail_factor, _outputs=("scen",))
        scen = out["scen"]
        scen.attrs["units"] = sim.attrs.get("units", "")  # "Detrending needs units." (adjustment.py:669-670)
        return scen


class QuantileDeltaMapping(EmpiricalQuantileMapping):
    """Quantile Delta Mapping bias-adjustment (adjustment.py:674-742; Cannon et al. 2015)."""

    def _adjust(self, sim, interp="nearest", extrapolation="constant", rank_window=None):
        out = qdm_adjust(self.ds.assign(sim=sim), group=self.group, interp=interp, extrapolation=extrapolation,
                         kind=self.kind, adapt_freq_thresh=self.adapt_freq_thresh, rank_window=rank_window,
                         max_tail_factor=self.max_tail_factor,
                         _outputs=("scen", "sim_q") if OPTIONS[EXTRA_OUTPUT] else ("scen",))  # (sim_q: float64, twice the size of sim)
        if OPTIONS[EXTRA_OUTPUT]:
            out["sim_q"].attrs.update(long_name="Group-wise quantiles of `sim`.")
            return out
        return out["scen"]




class Scaling(TrainAdjust):
    """Scaling bias-adjustment (adjustment.py:985-1050): an additive or multiplicative factor per group so that the mean
    of `hist` matches the mean of `ref`."""

    _allow_diff_calendars = False
    _allow_diff_training_times = False

    @classmethod
    def _train(cls, ref, hist, *, group="time", kind: str = ADDITIVE):
        from ._adjustment import scaling_train

        ds = scaling_train(Dataset({"ref": ref, "hist": hist}), group=group, kind=kind)
        ds["af"].attrs.update(long_name="Scaling adjustment factors")
        return ds, {"group": group, "kind": kind}

    def _adjust(self, sim, interp="nearest"):
        from ._adjustment import scaling_adjust

        return scaling_adjust(Dataset({"sim": sim, "af": self.ds["af"]}), group=self.group, interp=interp, kind=self.kind)["scen"]


class LOCI(TrainAdjust):
    """Local Intensity Scaling (adjustment.py:933-1009; Schmidli et al. 2006): wet-day intensity ratio above a threshold
    that is mapped from `ref` to `hist` through their CDFs."""

    _allow_diff_calendars = False
    _allow_diff_training_times = False

    @classmethod
    def _train(cls, ref, hist, *, thresh, group="time"):
        from ._adjustment import loci_train
        from .processing import quantity_to_float

        thresh = quantity_to_float(thresh, ref, "thresh")
        ds = loci_train(Dataset({"ref": ref, "hist": hist}), group=group, thresh=thresh)
        ds["af"].attrs.update(long_name="LOCI adjustment factors")
        ds["hist_thresh"].attrs.update(long_name="Threshold over modeled data")
        return ds, {"group": group, "thresh": thresh}

    def _adjust(self, sim, interp="linear"):
        from ._adjustment import loci_adjust

        return loci_adjust(Dataset({"hist_thresh": self.ds["hist_thresh"], "af": self.ds["af"], "sim": sim}), group=self.group,
                           thresh=self.thresh, interp=interp)["scen"]


for _cls in (EmpiricalQuantileMapping, DetrendedQuantileMapping, QuantileDeltaMapping, Scaling, LOCI):
    _register(_cls, "xsdba.adjustment")
