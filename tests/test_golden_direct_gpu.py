"""The reference-generated fixtures of the four kernels rows a3 / a11 / a12 / a18 stand on — `quantile_1d.npz` (nbutils.
_nan_quantile_1d on L in {1, 2, 30, ..., 3650}, NaN fractions, ties), `interp_1d.npz` (utils._interp_on_quantiles_1D: three
methods x two extrapolations, NaNs in xq / yq / newx), `interp_2d_nearest.npz` (utils._interp_on_quantiles_2d through griddata)
and `rank.npz` (scipy.stats.rankdata = bottleneck.nanrankdata) — fed to the HIP path DIRECTLY, through the C ABI: the device
meets the reference's own outputs without the oracle in between (tests/test_oracle_golden.py holds the oracle to the same
files).  Mirrors /root/reference/tests/test_nbutils.py:10-34 and tests/test_utils.py:68-217.

Tolerances: float64 quantiles, 1-D interpolation and ranks exact; float32 quantiles 2 ulp (the fixtures come from a plain-numpy
run of the reference source, which keeps the gamma >= 0.5 lerp branch in float32 where numba — and the device — promote it,
nbutils.py:104); 2-D nearest nodes exact except proven distance ties (cKDTree's choice is implementation-defined)."""
from __future__ import annotations

import numpy as np
import pytest

from conftest import assert_close, assert_close_or_tie

pytestmark = pytest.mark.gpu

F32_ULP2 = 2.5e-7


@pytest.fixture(scope="module")
def hip():
    from xsdba_amd import hip as h

    h.default_context(0)
    return h


@pytest.fixture(params=["cell", "rows"])
def kernels(request, monkeypatch):
    """both families: the per-cell kernels where eligible, and the generic ones pinned"""
    if request.param == "rows":
        monkeypatch.setenv("XSDBA_HIP_TRAIN", "rows")
        monkeypatch.setenv("XSDBA_HIP_QDM", "count")
    return request.param


def test_quantile_1d_fixture_on_device(hip, golden, kernels):
    d = golden("quantile_1d")
    n = int(d["n"])
    assert n > 100
    for i in range(n):
        arr, q, exp = d[f"arr_{i}"], d[f"q_{i}"], d[f"out_{i}"]
        got = hip.quantile_groups(arr[None, :], np.zeros(arr.size, np.int32), 1, 1, q)[0, 0]
        assert got.dtype == arr.dtype
        assert_close(got, exp, rtol=0.0 if arr.dtype == np.float64 else F32_ULP2, what=f"quantile case {i} L={arr.size} {arr.dtype}")
    # reference tests/test_nbutils.py:23-34: one valid value -> that value; none -> NaN
    q = np.linspace(0.1, 0.99, 50)
    one = np.array([[1.0] + [np.nan] * 100])
    got = hip.quantile_groups(one, np.zeros(101, np.int32), 1, 1, q)[0, 0]
    np.testing.assert_array_equal(got, d["edge_one_out"])
    got = hip.quantile_groups(np.full((1, 100), np.nan), np.zeros(100, np.int32), 1, 1, q)[0, 0]
    assert np.isnan(got).all() and np.isnan(d["edge_nan_out"]).all()
    # several of the fixture's rows as the groups of ONE series (ragged rows, the grouped kernels)
    rows = [d[f"arr_{i}"] for i in range(n) if d[f"arr_{i}"].dtype == np.float32 and d[f"q_{i}"].size == 20][:12]
    qs = d["q_0"]
    x = np.concatenate(rows)[None, :]
    g = np.concatenate([np.full(r.size, k, np.int32) for k, r in enumerate(rows)])
    got = hip.quantile_groups(x, g, len(rows), 1, qs)[0]
    k = 0
    for i in range(n):
        if d[f"arr_{i}"].dtype == np.float32 and d[f"q_{i}"].size == 20 and k < len(rows):
            assert_close(got[k], d[f"out_{i}"], rtol=F32_ULP2, what=f"grouped row {k} (case {i})")
            k += 1


def test_interp_1d_fixture_on_device(hip, golden):
    d = golden("interp_1d")
    n = int(d["n"])
    assert n >= 48
    for i in range(n):
        newx, xq, yq = d[f"newx_{i}"], d[f"xq_{i}"], d[f"yq_{i}"]
        method, extrap = str(d[f"method_{i}"]), str(d[f"extrap_{i}"])
        got = hip.qm_adjust(newx[None], yq[None, None], xq[None, None], None, method, extrap, kind=None)[0]
        rtol = 0.0 if method != "cubic" else (1e-11 if newx.dtype == np.float64 else 1e-5)  # (the spline solve: see test_interp_cubic_gpu.py)
        assert_close(got, d[f"out_{i}"], rtol=rtol, atol=0.0 if method != "cubic" else rtol * float(np.nanmax(np.abs(yq))),
                     what=f"interp1d case {i} {method}/{extrap} {newx.dtype}")


def test_interp_2d_nearest_fixture_on_device(hip, golden):
    d = golden("interp_2d_nearest")
    for i in range(int(d["n"])):
        sim, hq, af = d[f"sim_{i}"], d[f"hq_{i}"], d[f"af_{i}"]
        gidx = (d[f"newg_{i}"] - 1).astype(np.int32)
        got, nodes = hip.qm_adjust(sim, af, hq, gidx, "nearest", str(d[f"extrap_{i}"]), kind=None, return_nodes=True)
        assert_close_or_tie(got, d[f"out_{i}"], sim, hq, af, gidx, None, 0.0, what=f"interp2d case {i}")
        # the node indexes are an integer artefact: they must point at the value that came out
        # (flat index r * nq + k on the table wrapped with its cyclic rows: row 0 = the last group, row G + 1 = the first)
        ok = nodes >= 0
        G, Q = af.shape[1:]
        r, k = np.where(ok, nodes, 0) // Q, np.where(ok, nodes, 0) % Q
        g = np.where(r == 0, G - 1, np.where(r == G + 1, 0, r - 1))
        picked = af[np.arange(af.shape[0])[:, None], g, k]
        # (k counts the row's non-NaN pairs: checked on the cells whose tables hold no NaN)
        dense = ~(np.isnan(af).any(axis=(1, 2)) | np.isnan(hq).any(axis=(1, 2)))
        sel = ok & ~np.isnan(got) & dense[:, None]
        if not sel.any():
            continue
        np.testing.assert_array_equal(picked[sel], got[sel])


def test_rank_fixture_on_device(hip, golden, kernels):
    d = golden("rank")
    for i in range(int(d["n"])):
        arr = d[f"arr_{i}"]
        got = hip.rank_groups(arr, np.zeros(arr.shape[1], np.int32), 1, 1, False)
        assert got.dtype == np.float64
        assert_close(got, d[f"out_{i}"], rtol=0.0, atol=0.0, what=f"rank case {i}")
