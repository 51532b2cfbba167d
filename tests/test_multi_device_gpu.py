"""The node behind the drop-in API: `Cls.train(ref, hist, ...).adjust(sim)` on host cubes deals blocks of cells over one context
per device (XSDBA_AMD_DEVICES; default: every visible device), each context driven by its own thread — what dask gives the
reference (base.py:640-650, 715).  A one-GPU box rehearses it with TWO contexts on device 0: the results must equal the
one-context run bit for bit, and both contexts must have worked."""
from __future__ import annotations

import sys
import threading
import warnings

import numpy as np
import pytest

import xsdba_amd as xs
from xsdba_amd import _lib, hip
from xsdba_amd.labeled import DataArray

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    hip.default_context(0)


def _cube(values, dims, units):
    values = np.asarray(values)
    coords = {"time": xs.date_range("2000-01-01", values.shape[dims.index("time")], "noleap").values}
    for d, n in zip(dims, values.shape):
        if d != "time":
            coords[d] = np.arange(n) * 2.5
    return DataArray(values, dims, coords, {"units": units})


def _data(kind, shape, seed=5):
    rng = np.random.default_rng(seed)
    T = shape[-1]
    doy = np.arange(T) % 365
    if kind == "+":
        b = 280 - 20 * np.cos(2 * np.pi * (doy + 1) / 365)
        arrs = [b + 2 * rng.random(shape), b + 1 + 2.1 * rng.random(shape), b + 2 + 2.2 * rng.random(shape) + 0.1 * np.arange(T) / 365]
    else:
        arrs = [rng.gamma(4, 7.5, shape), rng.gamma(8.15, 3.68, shape), rng.gamma(16, 2.63, shape)]
    return [a.astype(np.float32) for a in arrs]


CASES = [
    ("EmpiricalQuantileMapping", dict(group=xs.Grouper("time.dayofyear", window=31), nquantiles=20), "+", {}),
    ("DetrendedQuantileMapping", dict(group=xs.Grouper("time.dayofyear", window=31), nquantiles=20), "+", {}),
    ("QuantileDeltaMapping", dict(group=xs.Grouper("time.dayofyear", window=31), nquantiles=20), "*", {}),
    ("EmpiricalQuantileMapping", dict(group="time.month", nquantiles=15), "*", {"interp": "linear"}),
]


@pytest.mark.parametrize("dims", [("lat", "lon", "time"), ("time", "lat", "lon")])
@pytest.mark.parametrize("min_mb", ["0", "1e12"])  # blocks uploaded through the device-resident path / the plain one
@pytest.mark.parametrize("clsname,train_kw,kind,adj_kw", CASES)
def test_two_contexts_equal_one(monkeypatch, clsname, train_kw, kind, adj_kw, dims, min_mb):
    cls = getattr(xs, clsname)
    units = "K" if kind == "+" else "mm/d"
    arrs = _data(kind, (8, 4, 3650))
    if dims[0] == "time":
        arrs = [np.ascontiguousarray(np.moveaxis(a, -1, 0)) for a in arrs]
    ref, hist, sim = (_cube(a, dims, units) for a in arrs)

    def run():
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            obj = cls.train(ref, hist, kind=kind, **train_kw)
            return obj, obj.adjust(sim, **adj_kw)

    monkeypatch.setenv("XSDBA_AMD_DEVICE_MIN_MB", min_mb)
    monkeypatch.setenv("XSDBA_AMD_MULTI_MIN_MB", "0")
    monkeypatch.setenv("XSDBA_AMD_DEVICES", "0")
    obj1, scen1 = run()
    monkeypatch.setenv("XSDBA_AMD_DEVICES", "0,0")
    seen, blocks = [], []
    real = _lib.use_context

    def spy(ctx):
        # (the block being run: the caller is base._map_blocks.one_block(b, ctx) — the two workers' calls arrive in any order)
        blk = sys._getframe(1).f_locals.get("b")
        seen.append((id(ctx), ctx.device, threading.current_thread().name))
        blocks.append((blk, id(ctx)))
        return real(ctx)

    monkeypatch.setattr(_lib, "use_context", spy)
    obj2, scen2 = run()
    ctxs = _lib.worker_contexts()
    assert len(ctxs) == 2 and ctxs[0] is not ctxs[1] and ctxs[0] is hip.default_context(0)
    assert {i for i, _, _ in seen} == {id(c) for c in ctxs}, seen              # both contexts worked ...
    assert len({t for _, _, t in seen}) == 2 and threading.current_thread().name not in {t for _, _, t in seen}  # ... each on its own thread
    assert len(seen) >= 8                                                       # two blocks per context for train, and for adjust
    if min_mb == "0" and dims[-1] == "time":  # (blocks of a (time, lat, lon) cube are strided host slices: they take the plain path)
        # the blocks' results stay on the device that computed them: the trained tables are sharded over the two contexts, and
        # every block of the adjustment ran on the context that holds its tables
        from xsdba_amd.labeled import ShardedDataArray

        af = obj2.ds["af"]
        assert isinstance(af, ShardedDataArray) and af.part_dim == "lat" and len(af.parts) == 4
        owners = [p.device_data.ctx for _, _, p in af.parts]
        assert {id(c) for c in owners} == {id(c) for c in ctxs}
        # (the calls behind the training's: whatever map_blocks pass of the adjustment they belong to, block b ran where part b lives)
        adj_calls = blocks[len(af.parts):]
        assert len(adj_calls) >= len(af.parts) and {b for b, _ in adj_calls} == set(range(len(af.parts)))
        assert all(i == id(owners[b]) for b, i in adj_calls), (adj_calls, owners)
    for name in obj1.ds.data_vars:
        a, b = obj1.ds[name], obj2.ds[name]
        assert a.dims == b.dims
        np.testing.assert_array_equal(np.asarray(a.values), np.asarray(b.values), err_msg=name)
    assert scen1.dims == scen2.dims
    np.testing.assert_array_equal(np.asarray(scen1.values), np.asarray(scen2.values))


def test_device_resident_inputs_stay_on_their_device(monkeypatch):
    """cubes that already live on a device are not cut up and shipped around: one call on the context that owns them"""
    monkeypatch.setenv("XSDBA_AMD_DEVICES", "0,0")
    monkeypatch.setenv("XSDBA_AMD_MULTI_MIN_MB", "0")
    ctx = hip.default_context(0)
    ref, hist, sim = (_cube(a, ("lat", "lon", "time"), "K") for a in _data("+", (4, 3, 1460)))
    dref, dhist, dsim = (DataArray(ctx.to_device(np.asarray(v.values)), v.dims, v.coords, v.attrs) for v in (ref, hist, sim))
    called = []
    real = _lib.use_context
    monkeypatch.setattr(_lib, "use_context", lambda c: (called.append(c), real(c))[1])
    obj = xs.EmpiricalQuantileMapping.train(dref, dhist, group=xs.Grouper("time.dayofyear", window=31), nquantiles=20)
    scen = obj.adjust(dsim)
    assert not called
    monkeypatch.setenv("XSDBA_AMD_DEVICES", "0")
    obj1 = xs.EmpiricalQuantileMapping.train(ref, hist, group=xs.Grouper("time.dayofyear", window=31), nquantiles=20)
    np.testing.assert_array_equal(np.asarray(scen.values), np.asarray(obj1.adjust(sim).values))
