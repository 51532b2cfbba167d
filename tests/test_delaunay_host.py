"""xsdba_amd/csrc/delaunay.hpp compiled for the host and checked against scipy (Qhull): the triangulation of table-like
node sets (rows of sorted abscissae at integer group coordinates, utils.py:476-513) and the values of
griddata(method="linear") (utils.py:391-396).  CPU-only; the GPU path is tested in tests/test_interp_linear_gpu.py."""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np
import pytest
from scipy.interpolate import CloughTocher2DInterpolator, griddata
from scipy.spatial import Delaunay

ROOT = Path(__file__).resolve().parents[1]


@pytest.fixture(scope="module")
def lib(tmp_path_factory):
    out = tmp_path_factory.mktemp("dt") / "libdt_host.so"
    subprocess.run(["g++", "-O2", "-shared", "-fPIC", "-o", str(out), str(ROOT / "tests/native/delaunay_host.cpp")], check=True)
    return C.CDLL(str(out))


def build(lib, px, py, row_order=0):
    n = len(px)
    tv = np.zeros(6 * n, np.uint16)
    tn = np.zeros(6 * n, np.uint16)
    nt = lib.dt_host_build(C.c_int(n), px.ctypes.data_as(C.c_void_p), py.ctypes.data_as(C.c_void_p), tv.ctypes.data_as(C.c_void_p),
                           tn.ctypes.data_as(C.c_void_p), C.c_int(row_order))
    return nt, tv, tn


def table_nodes(rng, rows, Q, spread=1.0, dtype=np.float32):
    """rows of Q sorted abscissae (float32-representable), group coordinate = row number"""
    xs_, ys_ = [], []
    for r in range(rows):
        x = np.sort((280 + 10 * np.cos(r / rows * 6.28) + spread * 8 * rng.standard_normal(Q)).astype(dtype)).astype(np.float64)
        xs_.append(np.unique(x))
        ys_.append(np.full(len(xs_[-1]), float(r)))
    return np.concatenate(xs_), np.concatenate(ys_)


@pytest.mark.parametrize("row_order", [0, 1])
@pytest.mark.parametrize("rows,Q,spread", [(14, 20, 1.0), (14, 50, 0.2), (6, 15, 3.0), (3, 4, 1.0), (40, 20, 1.0), (14, 1, 1.0), (2, 5, 1.0)])
def test_triangles_match_qhull(lib, rows, Q, spread, row_order):
    rng = np.random.default_rng(rows * 100 + Q)
    for rep in range(5):
        px, py = table_nodes(rng, rows, Q, spread)
        nt, tv, tn = build(lib, px, py, row_order)
        pts = np.c_[px, py]
        if rows >= 2 and len(px) >= 3 and len(np.unique(py)) > 1:
            ref = Delaunay(pts)
            mine = {tuple(sorted(tv[3 * t:3 * t + 3])) for t in range(nt)}
            theirs = {tuple(sorted(s)) for s in ref.simplices}
            # identical up to cocircular ties: compare through the interpolant instead when the sets differ
            if mine != theirs:
                assert len(mine) == len(theirs)
                assert len(mine ^ theirs) <= 0.02 * len(mine) + 2
            # counter-clockwise, consistent neighbours
            for t in range(nt):
                a, b, c = tv[3 * t:3 * t + 3]
                assert (px[b] - px[a]) * (py[c] - py[a]) - (py[b] - py[a]) * (px[c] - px[a]) > 0
                for k in range(3):
                    u = tn[3 * t + k]
                    if u != 0xFFFF:
                        assert t in tn[3 * u:3 * u + 3]
                        assert {tv[3 * t + (k + 1) % 3], tv[3 * t + (k + 2) % 3]} <= set(tv[3 * u:3 * u + 3])


def table_rows(rng, rows, Q, spread=1.0, drop=()):
    """table_nodes + the first node of every row (rows in `drop` are left empty)"""
    xs_, ys_, rowb = [], [], [0]
    for r in range(rows):
        x = np.sort((280 + 10 * np.cos(r / rows * 6.28) + spread * 8 * rng.standard_normal(Q)).astype(np.float32)).astype(np.float64)
        x = np.unique(x) if r not in drop else x[:0]
        xs_.append(x)
        ys_.append(np.full(len(x), float(r)))
        rowb.append(rowb[-1] + len(x))
    return np.concatenate(xs_), np.concatenate(ys_), np.array(rowb, np.int32)


@pytest.mark.parametrize("rows,Q,spread,drop", [(14, 20, 1.0, ()), (14, 50, 0.2, ()), (6, 15, 3.0, ()), (3, 4, 1.0, ()), (40, 20, 1.0, ()),
                                                 (2, 5, 1.0, ()), (14, 20, 0.02, ()), (14, 20, 30.0, ()), (14, 20, 1.0, (0, 5, 6, 13)),
                                                 (367, 20, 1.0, ()), (14, 2, 1.0, ())])
def test_row_construction_matches_qhull(lib, rows, Q, spread, drop):
    """the row-parallel construction (dtp_*: strips, pockets, rounds of arbitrated Lawson flips; its lanes emulated one after the
    other): Qhull's triangles, counter-clockwise, consistent neighbours; narrow and wide spreads (many / few flips), empty rows,
    day-of-year height, two nodes per row"""
    rng = np.random.default_rng(rows * 100 + Q)
    for rep in range(3):
        px, py, rowb = table_rows(rng, rows, Q, spread, drop)
        n = len(px)
        tv, tn, rounds = np.zeros(6 * n, np.uint16), np.zeros(6 * n, np.uint16), C.c_int(0)
        p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
        nt = lib.dt_host_build_rows(C.c_int(n), p(px), p(py), p(tv), p(tn), C.c_int(rows), p(rowb), C.byref(rounds))
        assert nt > 0 and rounds.value < 8 * n
        ref = Delaunay(np.c_[px, py])
        mine = {tuple(sorted(tv[3 * t:3 * t + 3])) for t in range(nt)}
        theirs = {tuple(sorted(s)) for s in ref.simplices}
        assert len(mine) == len(theirs) == nt
        assert len(mine ^ theirs) <= 0.02 * len(mine) + 2  # (cocircular ties)
        for t in range(nt):
            a, b, c = tv[3 * t:3 * t + 3]
            assert (px[b] - px[a]) * (py[c] - py[a]) - (py[b] - py[a]) * (px[c] - px[a]) > 0
            for k in range(3):
                u = tn[3 * t + k]
                if u != 0xFFFF:
                    assert t in tn[3 * u:3 * u + 3]
                    assert {tv[3 * t + (k + 1) % 3], tv[3 * t + (k + 2) % 3]} <= set(tv[3 * u:3 * u + 3])


def test_row_construction_declines_single_node_rows(lib):
    px = np.array([1.0, 2.0, 3.0, 1.5, 0.5, 2.5]); py = np.array([0.0, 0.0, 0.0, 1.0, 2.0, 2.0])
    rowb = np.array([0, 3, 4, 6], np.int32)
    tv, tn = np.zeros(36, np.uint16), np.zeros(36, np.uint16)
    p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    assert lib.dt_host_build_rows(C.c_int(6), p(px), p(py), p(tv), p(tn), C.c_int(3), p(rowb), None) == -1


@pytest.mark.parametrize("rows,Q", [(14, 20), (14, 50), (6, 15), (40, 20)])
def test_values_match_griddata_linear(lib, rows, Q):
    rng = np.random.default_rng(rows + Q)
    px, py = table_nodes(rng, rows, Q)
    val = rng.standard_normal(len(px))
    nt, tv, tn = build(lib, px, py)
    nq = 4000
    qx = (280 + 14 * rng.standard_normal(nq)).astype(np.float32).astype(np.float64)
    qy = rng.uniform(0.0, rows - 1.0, nq)
    qy[:200] = np.round(qy[:200])  # on the rows themselves
    out = np.empty(nq)
    lib.dt_host_eval(C.c_int(len(px)), px.ctypes.data_as(C.c_void_p), py.ctypes.data_as(C.c_void_p), val.ctypes.data_as(C.c_void_p),
                     C.c_int(nt), tv.ctypes.data_as(C.c_void_p), tn.ctypes.data_as(C.c_void_p), C.c_int(nq),
                     qx.ctypes.data_as(C.c_void_p), qy.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
    ref = griddata((px, py), val, (qx, qy), method="linear")
    both = np.isfinite(ref) & np.isfinite(out)
    assert (np.isfinite(ref) != np.isfinite(out)).mean() < 2e-3  # hull-boundary ties only
    np.testing.assert_allclose(out[both], ref[both], rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("rows,Q", [(14, 15), (14, 20), (6, 15), (40, 20)])
def test_values_match_griddata_cubic(lib, rows, Q):
    """griddata(method="cubic") = CloughTocher2DInterpolator(tol=1e-6, maxiter=400): nodal gradients (same Gauss-Seidel
    sweeps, same stopping sweep) and the values of the Clough-Tocher element"""
    rng = np.random.default_rng(rows + Q)
    px, py = table_nodes(rng, rows, Q)
    val = rng.standard_normal(len(px))
    nt, tv, tn = build(lib, px, py)
    nq = 4000
    qx = (280 + 14 * rng.standard_normal(nq)).astype(np.float32).astype(np.float64)
    qy = rng.uniform(0.0, rows - 1.0, nq)
    qy[:200] = np.round(qy[:200])
    out, grad = np.empty(nq), np.zeros(2 * len(px))
    p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    iters = lib.dt_host_eval_ct(C.c_int(len(px)), p(px), p(py), p(val), C.c_int(nt), p(tv), p(tn), C.c_int(nq), p(qx), p(qy), p(out), p(grad))
    assert 0 < iters < 400
    ct = CloughTocher2DInterpolator(np.c_[px, py], val)
    np.testing.assert_allclose(grad.reshape(-1, 2), ct.grad.reshape(-1, 2), rtol=1e-8, atol=1e-8)
    ref = griddata((px, py), val, (qx, qy), method="cubic")
    both = np.isfinite(ref) & np.isfinite(out)
    assert (np.isfinite(ref) != np.isfinite(out)).mean() < 2e-3
    np.testing.assert_allclose(out[both], ref[both], rtol=1e-8, atol=1e-8)
