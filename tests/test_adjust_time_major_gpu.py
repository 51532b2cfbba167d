"""qm_adjust on (time, cell) series resident on the device: the lane-per-cell kernel (xsdba_amd/csrc/adjust_tm.hpp) maps sim in
place and writes scen in the same layout.  Held bit for bit to the cell-major kernel (itself held to the oracle elsewhere)
and, on selected cases, to the oracle directly: day-of-year and monthly tables, NaN rows / NaN cells / NaN sim, nodes more
than one row step apart (the search leaves the element's own row), extrapolation "nan", a tile that is not full, node counts
that stage in 16-byte pieces (20, 32, 48) and that do not (15, 21)."""
from __future__ import annotations

import numpy as np
import pytest

import oracle
from xsdba_amd import hip

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    return hip.default_context(0)


def _tm(ctx, a):
    """(C, T) host array -> the (time, cell) device layout the library reads as a logical [cell, time] block"""
    d = ctx.to_device(np.ascontiguousarray(a.T))
    d.logical_shape, d.cell_time_strides = a.shape, (1, a.shape[0])
    return d


def _groups(name, T):
    d = np.arange(T) % 365
    if name == "doy":
        return d.astype(np.int32), 365
    cum = np.concatenate([[0], np.cumsum([31, 28, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31])])
    return (np.searchsorted(cum, d, side="right") - 1).astype(np.int32), 12


def _tables(rng, C, G, Q, kind, spread):
    base = (280 + 10 * rng.standard_normal((C, G, 1))) if kind == "+" else rng.gamma(2.0, 2.0, (C, G, 1))
    hq = np.sort(base + spread * rng.standard_normal((C, G, Q)).cumsum(-1) * (0.3 if kind == "+" else 0.5), axis=-1).astype(np.float32)
    af = (rng.standard_normal((C, G, Q)) if kind == "+" else rng.gamma(4.0, 0.25, (C, G, Q))).astype(np.float32)
    return hq, af


@pytest.mark.parametrize("gname,Q,kind,extrap,spread", [
    ("doy", 20, "+", "constant", 1.0), ("doy", 20, "*", "nan", 1.0), ("month", 20, "+", "constant", 1.0), ("doy", 15, "+", "constant", 1.0),
    ("month", 21, "*", "constant", 8.0), ("doy", 32, "+", "nan", 6.0), ("month", 48, "+", "constant", 1.0), ("doy", 20, "+", "constant", 12.0),
])
def test_time_major_adjust_equals_cell_major(ctx, gname, Q, kind, extrap, spread):
    import zlib

    rng = np.random.default_rng(zlib.crc32(f"{gname}-{Q}-{kind}-{extrap}-{spread}".encode()))
    C, T = 131, 3 * 365 + 17
    gidx, G = _groups(gname, T)
    hq, af = _tables(rng, C, G, Q, kind, spread)
    sim = (hq[:, gidx, :].mean(-1) + (3.0 if kind == "+" else 1.0) * spread * rng.standard_normal((C, T))).astype(np.float32)
    sim[rng.random(sim.shape) < 0.01] = np.nan
    sim[5] = np.nan                                  # a NaN series
    hq[7] = np.nan; af[7] = np.nan                   # a cell without a table (sim valid)
    af[9, 3, 2] = np.nan                             # a NaN factor: the pair is skipped
    hq[11, G // 2, :] = np.nan                       # a row without abscissae
    hq[13, :, Q - 1] = np.nan                        # the last node missing in every row
    want = hip.qm_adjust(sim, af, hq, gidx, "nearest", extrap, kind)
    d_sim, d_af, d_hq = _tm(ctx, sim), ctx.to_device(af), ctx.to_device(hq)
    got = hip.qm_adjust(d_sim, d_af, d_hq, gidx, "nearest", extrap, kind, ctx=ctx, out_time_major=True)
    assert got.shape == (T, C)
    np.testing.assert_array_equal(got.download().reshape(T, C).T, want)
    # a few cells straight against the oracle (the reference's algorithm on the CPU)
    sel = [0, 7, 9, 11, 13, C - 1]
    exp = oracle.qm_adjust(sim[sel], af[sel], hq[sel], gidx, "nearest", extrap, kind)
    np.testing.assert_array_equal(got.download().reshape(T, C).T[sel], exp)


def test_time_major_adjust_runs_the_lane_kernel(ctx):
    """the (time, cell) call is served by k_qm_adjust_tm — no transposes, no k_qm_adjust"""
    rng = np.random.default_rng(3)
    C, T, Q = 64, 730, 20
    gidx, G = _groups("doy", T)
    hq, af = _tables(rng, C, G, Q, "+", 1.0)
    sim = (hq[:, gidx, :].mean(-1) + rng.standard_normal((C, T))).astype(np.float32)
    d_sim, d_af, d_hq = _tm(ctx, sim), ctx.to_device(af), ctx.to_device(hq)
    ctx.profile_enable(True)
    ctx.profile_read(reset=True)
    hip.qm_adjust(d_sim, d_af, d_hq, gidx, "nearest", "constant", "+", ctx=ctx, out_time_major=True)
    ctx.synchronize()
    prof = {k: v for k, v in ctx.profile_read(reset=True).items() if v["count"]}
    ctx.profile_enable(False)
    assert "k_qm_adjust_tm" in prof and not any(k.startswith("k_transpose") or k == "k_qm_adjust" for k in prof), prof


def test_api_returns_time_major_scen_for_time_major_sim(ctx, monkeypatch):
    """EmpiricalQuantileMapping.adjust on a device-resident (time, lat, lon) sim: scen comes back (time, lat, lon), equal to the
    host path's"""
    import warnings

    import xsdba_amd as xs
    from xsdba_amd.labeled import DataArray

    rng = np.random.default_rng(11)
    shape = (4, 5, 3 * 365)
    doy = np.arange(shape[-1]) % 365
    base = 280 - 20 * np.cos(2 * np.pi * (doy + 1) / 365)
    ref, hist, sim = [(base + k + 2 * rng.random(shape)).astype(np.float32) for k in range(3)]
    t = xs.date_range("2000-01-01", shape[-1], "noleap").values

    def cube(v, dims):
        return DataArray(v, dims, {"time": t, "lat": np.arange(4), "lon": np.arange(5)}, {"units": "K"})

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        monkeypatch.setenv("XSDBA_AMD_DEVICE_MIN_MB", "1e12")
        obj = xs.EmpiricalQuantileMapping.train(cube(ref, ("lat", "lon", "time")), cube(hist, ("lat", "lon", "time")),
                                                group=xs.Grouper("time.dayofyear", window=31), nquantiles=20)
        want = obj.adjust(cube(sim, ("lat", "lon", "time")))
        sim_tm = np.ascontiguousarray(sim.transpose(2, 0, 1))
        d = ctx.to_device(sim_tm)
        got = obj.adjust(DataArray(d, ("time", "lat", "lon"), {"time": t, "lat": np.arange(4), "lon": np.arange(5)}, {"units": "K"}))
    # the labels follow map_blocks' template order (like the host path's); the BUFFER is sim's layout: no layout pass either way
    assert got.dims == want.dims and got.device_data is not None and got.device_dims == ("time", "lat", "lon")
    np.testing.assert_array_equal(got.values, want.values)
