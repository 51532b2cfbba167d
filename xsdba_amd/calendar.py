"""Time-axis helpers: calendar fields (year, month, day, dayofyear, days_in_month) of a time coordinate.

The reference gets these from xarray/pandas/cftime indexes (``ind.month``, ``ind.dayofyear``,
``da.time.dt.calendar`` — base.py:306-332, adjustment.py:105-108).  Neither xarray nor cftime is a
dependency of this package, so a small :class:`TimeIndex` carries the same fields; it is built from
``numpy.datetime64`` arrays (standard calendar), from anything exposing ``year/month/day`` per element
(cftime objects, duck-typed), or generated with :func:`date_range` for the CF calendars ``noleap``
/``365_day``, ``360_day``, ``all_leap``/``366_day`` and ``standard``.
"""
from __future__ import annotations

import numpy as np

_DPM = {
    "noleap": np.array([31, 28, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31]),
    "all_leap": np.array([31, 29, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31]),
    "360_day": np.full(12, 30),
}
_ALIASES = {"365_day": "noleap", "366_day": "all_leap", "gregorian": "standard", "proleptic_gregorian": "standard",
            "default": "standard"}

# base.py:105-115
MAX_DOY = {"standard": 366, "gregorian": 366, "proleptic_gregorian": 366, "julian": 366, "noleap": 365, "365_day": 365,
           "all_leap": 366, "366_day": 366, "360_day": 360, "default": 366}


def _canon(cal: str) -> str:
    return _ALIASES.get(cal, cal)


def _is_leap(year):
    return (year % 4 == 0) & ((year % 100 != 0) | (year % 400 == 0))


class TimeIndex:
    """Calendar fields of a 1-D time axis + a float64 numeric axis (ns since 1970-01-01 in its calendar,
    the abscissa xarray's polyfit uses, detrending.py:202-203)."""

    def __init__(self, year, month, day, calendar: str = "standard", values=None, frac=None):
        if _canon(calendar) == "julian":
            raise NotImplementedError("the 'julian' calendar is not supported (standard / proleptic_gregorian, noleap, all_leap, 360_day are)")
        self.year = np.asarray(year, np.int64)
        self.month = np.asarray(month, np.int64)
        self.day = np.asarray(day, np.int64)
        self.calendar = calendar
        self._values = values
        # time of day as a fraction of the day (non-standard calendars: datetime64 values carry it themselves); None = midnight
        self.frac = None if frac is None or not np.any(frac) else np.asarray(frac, np.float64)

    def __len__(self):
        return self.year.size

    @property
    def size(self):
        return self.year.size

    @property
    def values(self):
        """what the labelled layer stores as the coordinate: datetime64 for the standard calendar, else self"""
        return self._values if self._values is not None else self

    # ---- derived fields -------------------------------------------------------------------------
    def _dpm_table(self):
        cal = _canon(self.calendar)
        if cal == "standard":
            tab = np.tile(_DPM["noleap"], (self.size, 1))
            tab[:, 1] += _is_leap(self.year).astype(np.int64)
            return tab
        return np.tile(_DPM[cal], (self.size, 1))

    @property
    def days_in_month(self):
        return self._dpm_table()[np.arange(self.size), self.month - 1]

    @property
    def dayofyear(self):
        tab = self._dpm_table()
        cum = np.concatenate([np.zeros((self.size, 1), np.int64), np.cumsum(tab, axis=1)], axis=1)
        return cum[np.arange(self.size), self.month - 1] + self.day

    @property
    def week(self):
        """ISO 8601 week number (1..53), what pandas' ``isocalendar().week`` gives (base.py:323-325); standard calendar only,
        like the reference (xarray has no week accessor for cftime axes)"""
        if _canon(self.calendar) != "standard":
            raise NotImplementedError("week grouping needs the standard calendar (no ISO weeks on a cftime axis, as in the reference)")
        d = np.array([np.datetime64(f"{y:04d}-{m:02d}-{dd:02d}") for y, m, dd in zip(self.year, self.month, self.day)]).astype("datetime64[D]")
        dow = (d.astype(np.int64) + 3) % 7  # Monday = 0 (1970-01-01 was a Thursday)
        thursday = d + (3 - dow).astype("timedelta64[D]")  # the Thursday of the same ISO week decides the ISO year
        ystart = thursday.astype("datetime64[Y]").astype("datetime64[D]")
        return ((thursday - ystart).astype(np.int64) // 7 + 1).astype(np.int64)

    @property
    def is_leap_year(self):
        cal = _canon(self.calendar)
        if cal == "standard":
            return _is_leap(self.year)
        return np.full(self.size, cal == "all_leap")

    @property
    def days_since_1970(self):
        cal = _canon(self.calendar)
        if cal == "standard":
            d = np.array([np.datetime64(f"{y:04d}-{m:02d}-{dd:02d}") for y, m, dd in zip(self.year, self.month, self.day)])
            return (d - np.datetime64("1970-01-01")).astype("timedelta64[D]").astype(np.int64)
        ylen = {"noleap": 365, "all_leap": 366, "360_day": 360}[cal]
        return (self.year - 1970) * ylen + self.dayofyear - 1

    @property
    def numeric_ns(self) -> np.ndarray:
        if self._values is not None and np.issubdtype(np.asarray(self._values).dtype, np.datetime64):
            return np.asarray(self._values).astype("datetime64[ns]").astype(np.int64).astype(np.float64)
        days = self.days_since_1970.astype(np.float64)
        if self.frac is not None:
            days = days + self.frac
        return days * 86400e9

    def equals(self, other: "TimeIndex") -> bool:
        return (self.size == other.size and _canon(self.calendar) == _canon(other.calendar)
                and np.array_equal(self.year, other.year) and np.array_equal(self.month, other.month)
                and np.array_equal(self.day, other.day))

    def __getitem__(self, key):
        v = None if self._values is None else np.asarray(self._values)[key]
        return TimeIndex(self.year[key], self.month[key], self.day[key], self.calendar, v, None if self.frac is None else self.frac[key])

    def __repr__(self):
        if self.size == 0:
            return f"TimeIndex([], calendar={self.calendar!r})"
        f = lambda i: f"{self.year[i]:04d}-{self.month[i]:02d}-{self.day[i]:02d}"  # noqa: E731
        return f"TimeIndex({f(0)} .. {f(-1)}, n={self.size}, calendar={self.calendar!r})"


def date_range(start: str, periods: int, calendar: str = "standard") -> TimeIndex:
    """Daily time axis of `periods` days from `start` ("YYYY-MM-DD") in a CF calendar."""
    cal = _canon(calendar)
    if cal == "standard":
        vals = np.datetime64(start, "D") + np.arange(periods)
        return time_index(vals)
    y0, m0, d0 = (int(s) for s in start.split("-"))
    dpm = _DPM[cal]
    ylen = int(dpm.sum())
    doy0 = int(dpm[: m0 - 1].sum()) + d0 - 1
    n = doy0 + np.arange(periods)
    year = y0 + n // ylen
    d = n % ylen
    cum = np.concatenate([[0], np.cumsum(dpm)])
    month = np.searchsorted(cum, d, side="right")
    day = d - cum[month - 1] + 1
    return TimeIndex(year, month, day, calendar)


def time_index(values, calendar: str | None = None) -> TimeIndex:
    """Build a :class:`TimeIndex` from a time coordinate (datetime64 array, TimeIndex, pandas index,
    or an object array of cftime-like elements with ``year/month/day`` and optionally ``calendar``)."""
    if isinstance(values, TimeIndex):
        return values
    if hasattr(values, "year") and hasattr(values, "month") and hasattr(values, "day") and not isinstance(values, np.ndarray):
        # pandas DatetimeIndex / xarray CFTimeIndex
        cal = calendar or getattr(values, "calendar", "standard")
        raw = getattr(values, "values", None)
        raw = raw if raw is not None and np.issubdtype(np.asarray(raw).dtype, np.datetime64) else None
        return TimeIndex(np.asarray(values.year), np.asarray(values.month), np.asarray(values.day), cal, raw)
    arr = np.asarray(values)
    if np.issubdtype(arr.dtype, np.datetime64):
        d = arr.astype("datetime64[D]")
        y = d.astype("datetime64[Y]")
        m = d.astype("datetime64[M]")
        year = y.astype(np.int64) + 1970
        month = (m - y.astype("datetime64[M]")).astype(np.int64) + 1
        day = (d - m.astype("datetime64[D]")).astype(np.int64) + 1
        return TimeIndex(year, month, day, calendar or "standard", arr)
    if arr.dtype == object and arr.size and hasattr(arr.flat[0], "year"):
        cal = calendar or getattr(arr.flat[0], "calendar", "standard")
        return TimeIndex([v.year for v in arr.flat], [v.month for v in arr.flat], [v.day for v in arr.flat], cal)
    raise TypeError(f"cannot interpret a time coordinate of dtype {arr.dtype}")
