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
    dim, step = base._plan_blocks(ds, ["lat", "lon", "time"], keep_whole=("time",), n_workers=4)
    assert dim == "lat" and -(-40 // step) >= 8               # never the grouping dimension; two blocks per device
    assert sum(min(step, 40 - i0) for i0 in range(0, 40, step)) == 40
    # too small to be worth a second device
    monkeypatch.setenv("XSDBA_AMD_MULTI_MIN_MB", "64")
    assert base._plan_blocks(ds, ["lat", "lon", "time"], keep_whole=("time",), n_workers=4) is None
    # a budget smaller than a device's share still rules
    monkeypatch.setenv("XSDBA_AMD_MULTI_MIN_MB", "0")
    monkeypatch.setenv("XSDBA_AMD_BLOCK_MAX_MB", str(2 * 40 * 30 * 365 * 4 / 2 ** 20 / 20))
    dim, step = base._plan_blocks(ds, ["lat", "lon", "time"], keep_whole=("time",), n_workers=2)
    assert dim == "lat" and step == 2


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
