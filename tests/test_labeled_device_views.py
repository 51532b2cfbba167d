"""Lazily transposed device-resident arrays (labeled.DataArray.transpose relabels, the buffer keeps its layout):
`utils.cell_time_view` must order the cells the way the LABELS do, because callers pair the cells of several arrays by
position after `x.transpose(*other.dims)` (_adjustment._prep_train, processing.adapt_freq, detrending._apply).
CPU test with a stand-in for `_lib.DeviceArray` (same attributes; no GPU)."""
import numpy as np

from xsdba_amd.labeled import DataArray
from xsdba_amd.utils import cell_time_view


class DeviceArray:  # the name is what labeled._is_device_array looks at
    def __init__(self, host):
        self._h = np.ascontiguousarray(host)
        self.shape, self.dtype, self.ptr, self.ctx = self._h.shape, self._h.dtype, 1, None

    def download(self):
        return self._h.reshape(-1).copy()

    def reshape(self, *shape):
        shape = shape[0] if len(shape) == 1 and isinstance(shape[0], tuple) else shape
        return DeviceArray(self._h.reshape(shape))


def _rows(v):
    return v._h if isinstance(v, DeviceArray) else np.asarray(v)


def test_swapped_cell_order_is_paired_by_label():
    rng = np.random.default_rng(0)
    a = rng.random((2, 3, 5)).astype(np.float32)                       # (lat, lon, time)
    ref = DataArray(DeviceArray(a), ("lat", "lon", "time"))
    hist = DataArray(DeviceArray(a.transpose(1, 0, 2)), ("lon", "lat", "time"))  # the same cells stored (lon, lat, time)
    r2, cd_r, cs_r = cell_time_view(ref, "time")
    h2, cd_h, cs_h = cell_time_view(hist.transpose(*ref.dims), "time")
    assert cd_r == cd_h == ("lat", "lon") and tuple(cs_r) == tuple(cs_h) == (2, 3)
    np.testing.assert_array_equal(_rows(r2), _rows(h2))
    assert isinstance(r2, DeviceArray)  # the matching layout still takes the no-copy view


def test_time_in_the_middle_of_a_relabelled_buffer():
    a = np.arange(24, dtype=np.float32).reshape(2, 4, 3)                # buffer (lat, time, lon)
    da = DataArray(DeviceArray(a), ("lat", "time", "lon")).transpose("time", "lat", "lon")
    v, cd, cs = cell_time_view(da, "time")
    assert cd == ("lat", "lon") and tuple(cs) == (2, 3) and _rows(v).shape == (6, 4)
    np.testing.assert_array_equal(_rows(v), a.transpose(0, 2, 1).reshape(6, 4))


def test_result_layout_keeps_the_view():
    a = np.arange(24, dtype=np.float32).reshape(2, 3, 4)                # buffer (lat, lon, time), labelled like the template
    da = DataArray(DeviceArray(a), ("lat", "lon", "time")).transpose("time", "lat", "lon")
    v, cd, cs = cell_time_view(da, "time")
    assert isinstance(v, DeviceArray) and cd == ("lat", "lon") and tuple(cs) == (2, 3)
    np.testing.assert_array_equal(_rows(v), a.reshape(6, 4))


def test_rename_of_a_lazy_view():
    a = np.arange(24, dtype=np.float32).reshape(2, 3, 4)
    da = DataArray(DeviceArray(a), ("lat", "lon", "time")).transpose("time", "lat", "lon").rename({"lat": "y"})
    assert da.dims == ("time", "y", "lon") and da.shape == (4, 2, 3) and dict(da.sizes) == {"time": 4, "y": 2, "lon": 3}
    np.testing.assert_array_equal(da.values, a.transpose(2, 0, 1))
