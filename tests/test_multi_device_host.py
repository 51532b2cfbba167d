"""The node behind the drop-in API (the reference gets it from dask: base.py:640-650, 715): host-backed inputs are cut in
blocks of cells and dealt over one context per visible device.  CPU side: which devices, how the blocks are planned, and that
the random seeds of the blocks do not depend on the order their threads run in."""
from __future__ import annotations

import threading

import numpy as np
import pytest

import xsdba_amd as xs
from xsdba_amd import _lib, base
from xsdba_amd.labeled import DataArray, Dataset


def _cube(shape, dims):
    values = np.zeros(shape, np.float32)
    coords = {"time": xs.date_range("2000-01-01", shape[dims.index("time")], "noleap").values}
    return DataArray(values, dims, coords, {"units": "K"})


def test_worker_devices(monkeypatch):
    monkeypatch.setattr(_lib, "device_count", lambda: 8)
    for var in ("XSDBA_AMD_DEVICES", "LOCAL_RANK", "XSDBA_AMD_DEVICE"):
        monkeypatch.delenv(var, raising=False)
    assert _lib.worker_devices() == list(range(8))          # every visible device, no user code
    monkeypatch.setenv("LOCAL_RANK", "3")                     # one process per GPU (torchrun): the process keeps to its device
    assert _lib.worker_devices() == [3]
    monkeypatch.delenv("LOCAL_RANK")
    monkeypatch.setenv("XSDBA_AMD_DEVICE", "5")
    assert _lib.worker_devices() == [5]
    monkeypatch.setenv("XSDBA_AMD_DEVICES", "all")            # ... unless told otherwise
    assert _lib.worker_devices() == list(range(8))
    monkeypatch.setenv("XSDBA_AMD_DEVICES", "0, 2,2")         # an ordinal may repeat: two contexts on one device
    assert _lib.worker_devices() == [0, 2, 2]
    monkeypatch.setenv("XSDBA_AMD_DEVICES", "9")
    with pytest.raises(ValueError):
        _lib.worker_devices()


@pytest.mark.parametrize("dims,shape", [(("lat", "lon", "time"), (40, 30, 365)), (("time", "lat", "lon"), (365, 40, 30))])
def test_plan_blocks_shares_cells_over_workers(monkeypatch, dims, shape):
    monkeypatch.setenv("XSDBA_AMD_MULTI_MIN_MB", "0")
    ds = Dataset({"ref": _cube(shape, dims), "hist": _cube(shape, dims)})
    assert base._plan_blocks(ds, ["lat", "lon", "time"], keep_whole=("time",), n_workers=1) is None
    dim, bounds, ctxs, keep = base._plan_blocks(ds, ["lat", "lon", "time"], keep_whole=("time",), n_workers=4)
    assert dim == "lat" and len(bounds) >= 8                   # never the grouping dimension; two blocks per device
    assert bounds[0][0] == 0 and bounds[-1][1] == 40 and all(a[1] == b[0] for a, b in zip(bounds, bounds[1:]))
    assert ctxs is None and keep                               # dealt round-robin; the results stay on their devices
    # too small to be worth a second device
    monkeypatch.setenv("XSDBA_AMD_MULTI_MIN_MB", "64")
    assert base._plan_blocks(ds, ["lat", "lon", "time"], keep_whole=("time",), n_workers=4) is None
    # a budget smaller than a device's share still rules
    monkeypatch.setenv("XSDBA_AMD_MULTI_MIN_MB", "0")
    monkeypatch.setenv("XSDBA_AMD_BLOCK_MAX_MB", str(2 * 40 * 30 * 365 * 4 / 2 ** 20 / 20))
    dim, bounds, ctxs, keep = base._plan_blocks(ds, ["lat", "lon", "time"], keep_whole=("time",), n_workers=2)
    assert dim == "lat" and bounds[0] == (0, 2) and len(bounds) == 20 and not keep   # (over the budget: results come back to the host)


def test_sharded_array_is_lazy_and_plans_its_own_blocks():
    """labeled.ShardedDataArray: per-block parts laid end to end; `values` gathers, one part is handed over untouched, relabelling
    is lazy; a dataset holding one is cut along ITS parts."""
    from xsdba_amd.labeled import ShardedDataArray

    a = np.arange(40 * 6, dtype=np.float32).reshape(40, 6)
    parts = [DataArray(a[i:j], ("lat", "q"), {"lat": np.arange(i, j)}) for i, j in ((0, 16), (16, 30), (30, 40))]
    sh = ShardedDataArray(parts, "lat", name="af")
    assert sh.shape == (40, 6) and sh.dtype == np.float32 and sh.device_data is None
    assert sh.isel(lat=slice(16, 30)).values.base is parts[1].values.base or np.shares_memory(sh.isel(lat=slice(16, 30)).values, a)
    np.testing.assert_array_equal(sh.coords["lat"].values, np.arange(40))
    np.testing.assert_array_equal(sh.transpose("q", "lat").values, a.T)
    assert sh.rename({"lat": "y"}).part_dim == "y"
    np.testing.assert_array_equal(sh.isel(lat=slice(3, 20)).values, a[3:20])     # not a part: from the gathered copy
    np.testing.assert_array_equal(sh.values, a)
    ds = Dataset({"af": ShardedDataArray(parts, "lat", name="af"), "sim": _cube((40, 365), ("lat", "time"))})
    assert isinstance(ds["af"], ShardedDataArray)
    # parts without a device behind them are no reason to pin blocks to contexts: the ordinary plan applies
    assert base._plan_blocks(ds, ["lat", "time"], keep_whole=("time",), n_workers=1) is None


def test_block_seeds_do_not_depend_on_thread_order():
    from xsdba_amd.processing import _draw_seed

    def run(order):
        np.random.seed(7)
        shared = {"lock": threading.Lock(), "base": None}
        got = {}

        def work(block):
            _lib._tls.seeds = {"shared": shared, "block": block, "k": 0}
            try:
                got[block] = [_draw_seed() for _ in range(3)]
            finally:
                _lib._tls.seeds = None

        for b in order:
            th = threading.Thread(target=work, args=(b,))
            th.start()
            th.join()
        return got

    a, b = run([0, 1, 2, 3]), run([3, 1, 0, 2])
    assert a == b
    assert len({s for v in a.values() for s in v}) == 12     # distinct seeds for distinct (block, draw)
    np.random.seed(7)
    plain = _draw_seed()
    assert plain != a[0][0] and 0 <= a[0][0] < 2 ** 63
