"""Shared pytest configuration: markers, repo path, golden-vector loader."""
from __future__ import annotations

import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = Path(__file__).resolve().parent / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running")


@pytest.fixture(scope="session")
def golden():
    def _load(name):
        return np.load(GOLDEN / f"{name}.npz")

    return _load


def assert_close(a, b, rtol, atol=0.0, what=""):
    """NaN-pattern must match exactly; finite values within |a-b| <= atol + rtol*|b|."""
    a = np.asarray(a)
    b = np.asarray(b)
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    na, nb = np.isnan(a), np.isnan(b)
    assert np.array_equal(na, nb), f"{what}: NaN pattern differs at {np.argwhere(na != nb)[:5].tolist()}"
    ok = ~na
    ia, ib = np.isinf(a) & ok, np.isinf(b) & ok
    assert np.array_equal(ia, ib) and np.array_equal(a[ia], b[ib]), f"{what}: inf pattern differs"
    ok &= ~ia
    err = np.abs(a[ok].astype(np.float64) - b[ok].astype(np.float64))
    tol = atol + rtol * np.abs(b[ok].astype(np.float64))
    bad = err > tol
    assert not bad.any(), f"{what}: {bad.sum()} of {bad.size} beyond tol; max err {err.max():.3e} (tol {tol[err.argmax()]:.3e})"


def nearest_tie_values(x, r, hq_cell, af_cell):
    """All table values whose node is at the minimal 2-D distance from (x, r) on the cyclically
    extended table — more than one only at an exact distance tie (implementation-defined in cKDTree,
    SURVEY §9.10; common for doy/window tables where rows r-1 and r+1 share order statistics)."""
    G = hq_cell.shape[0]
    ext = np.concatenate([hq_cell[-1:], hq_cell, hq_cell[:1]]).astype(np.float64)
    aext = np.concatenate([af_cell[-1:], af_cell, af_cell[:1]])
    rows = np.arange(G + 2, dtype=np.float64)[:, None]
    d2 = (ext - np.float64(x)) ** 2 + (rows - r) ** 2
    d2[np.isnan(aext)] = np.nan
    return aext[d2 == np.nanmin(d2)]


def assert_close_or_tie(got, exp, sim, hq, af, gidx, kind, rtol, what="", newx=None):
    """assert_close, except that points where the reference picked another node of an exact
    nearest-neighbour tie are accepted when `got` corresponds to one of the tied nodes.
    `newx` is the interpolation abscissa when it is not `sim` itself (QDM: sim_q); `hq` may be a
    single [Q] row shared by all cells and groups (QDM: the quantile nodes)."""
    got = np.asarray(got)
    exp = np.asarray(exp)
    assert np.array_equal(np.isnan(got), np.isnan(exp)), f"{what}: NaN pattern differs"
    ok = ~np.isnan(exp)
    bad = ok & (np.abs(got.astype(np.float64) - exp) > rtol * np.abs(exp))
    for c, t in np.argwhere(bad):
        x = sim[c, t]
        xi = x if newx is None else newx[c, t]
        hq_c = hq[c] if np.ndim(hq) == 3 else np.broadcast_to(hq, af[c].shape)
        cands = nearest_tie_values(xi, gidx[t] + 1, hq_c, af[c])
        assert cands.size > 1, f"{what}: mismatch at {(c, t)} without a distance tie"
        vals = cands if kind is None else (x + cands if kind == "+" else x * cands)
        assert np.any(np.abs(vals.astype(np.float64) - got[c, t]) <= rtol * np.abs(got[c, t]) + 1e-30), \
            f"{what}: value at {(c, t)} is not one of the tied nearest nodes"
    assert bad.sum() <= max(5, 1e-2 * bad.size), f"{what}: too many tie points ({bad.sum()})"
