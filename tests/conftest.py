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


def nearest_tie_values(x, r, hq_cell, af_cell, delta=0.0):
    """All table values whose node is at the minimal 2-D distance from (x, r) on the cyclically
    extended table — more than one only at an exact distance tie (implementation-defined in cKDTree,
    SURVEY §9.10; common for doy/window tables where rows r-1 and r+1 share order statistics).
    delta > 0: nodes whose distance is within delta of the minimum count as tied (a knife-edge: an abscissa that differs
    by a few ulps may pick either)."""
    G = hq_cell.shape[0]
    if G == 1:
        ext, aext, rows, r = hq_cell.astype(np.float64), af_cell, np.zeros((1, 1)), 0.0
    else:
        ext = np.concatenate([hq_cell[-1:], hq_cell, hq_cell[:1]]).astype(np.float64)
        aext = np.concatenate([af_cell[-1:], af_cell, af_cell[:1]])
        rows = np.arange(G + 2, dtype=np.float64)[:, None]
    d2 = (ext - np.float64(x)) ** 2 + (rows - r) ** 2
    d2[np.isnan(aext)] = np.nan
    dmin = np.nanmin(d2)
    return aext[d2 <= (np.sqrt(dmin) + delta) ** 2]


def assert_close_or_tie(got, exp, sim, hq, af, gidx, kind, rtol, what="", newx=None, max_frac=1e-3):
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
    # every accepted point above is a proven tie; their number stays small on continuous data (a handful per 10^4 points)
    # (QDM passes a larger max_frac: its abscissae are the rationals k/(N-1), which hit the midpoint of two float32 nodes
    #  exactly for a few ranks of every group — structural, each one proven above)
    assert bad.sum() <= max(8, max_frac * bad.size), f"{what}: too many tie points ({bad.sum()} of {bad.size})"


def assert_dqm_close_or_flip(got, exp, sim, scaling, trend, hq, af, gidx, kind, rtol, what="", max_flips=8):
    """DQM adjust (reference _adjustment.py:679-780): scen = retrend(qm(detrend(scale(sim)))).  The trend may differ from
    the expected one in its last bit, which moves the detrended value by an ulp and — when that value sits on the knife-edge
    between two table nodes — selects the neighbouring node.  Points beyond `rtol` are accepted ONLY when that is proven:
    the detrended abscissa (recomputed here in float64) has more than one node within 4 float32 ulps (of the series' magnitude)
    of the minimal distance, and `got` equals the formula evaluated with one of those nodes.  At most `max_flips` points."""
    got = np.asarray(got)
    exp = np.asarray(exp)
    assert np.array_equal(np.isnan(got), np.isnan(exp)), f"{what}: NaN pattern differs"
    ok = ~np.isnan(exp)
    bad = ok & (np.abs(got.astype(np.float64) - exp) > rtol * np.abs(exp))
    G = hq.shape[1]
    for c, t in np.argwhere(bad):
        g = int(gidx[t]) if G > 1 else 0
        s_ = np.float64(scaling[c, g])
        xs = np.float64(sim[c, t]) + s_ if kind == "+" else np.float64(sim[c, t]) * s_
        tr = np.float64(trend[c, t])
        xd = xs - tr if kind == "+" else xs / tr
        delta = 4 * 1.1920929e-07 * max(abs(xs), abs(tr), abs(xd))
        cands = nearest_tie_values(xd, g + 1, hq[c], af[c], delta=delta)
        assert cands.size > 1, f"{what}: mismatch at {(c, t)} (got {got[c, t]}, expected {exp[c, t]}) without a knife-edge"
        vals = (xd + cands.astype(np.float64)) + tr if kind == "+" else (xd * cands.astype(np.float64)) * tr
        assert np.any(np.abs(vals - got[c, t]) <= 10 * rtol * np.abs(got[c, t]) + 1e-30), \
            f"{what}: value at {(c, t)} is not the formula at one of the equidistant nodes"
    assert bad.sum() <= max_flips, f"{what}: {bad.sum()} knife-edge points (more than {max_flips})"
