"""The grouped interp="linear" / "cubic" kernels of round 4 against those of rounds 1-3 and against scipy.griddata:
  * k_lin2d_build_rows (a workgroup per cell: row strips, pockets, arbitrated parallel Lawson flips; in LDS for small tables, in
    global memory for day-of-year ones) vs the one-thread sweep (XSDBA_HIP_LIN2D_BUILD=sweep);
  * the strip lists (a bisection over the edges crossing the query's strip) vs the walk (XSDBA_HIP_LIN2D_WALK=1).
The Delaunay triangulation is unique up to cocircular ties and the interpolant is continuous, so all four combinations give the
same values up to rounding.  Tables with NaN nodes (shorter rows, an empty row, a row of one node: the sweep fallback inside the
new kernel), queries on the rows and on the nodes themselves, beyond both ends of the rows, a float64 table and row coordinates
float32 cannot hold (both take the walk by themselves), a day-of-year table."""
from __future__ import annotations

import numpy as np
import pytest
from scipy.interpolate import griddata

from xsdba_amd import hip

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    hip.default_context(0)


def _table(rng, C, G, Q, dtype=np.float32):
    hq = np.sort(280 + 6 * np.cos(np.arange(G)[None, :, None] / G * 6.28) + 8 * rng.standard_normal((C, G, Q)), axis=2).astype(dtype)
    af = rng.standard_normal((C, G, Q)).astype(dtype)
    return af, hq


def _run(monkeypatch, build, walk, *args, **kw):
    monkeypatch.delenv("XSDBA_HIP_LIN2D_BUILD", raising=False)
    monkeypatch.delenv("XSDBA_HIP_LIN2D_WALK", raising=False)
    if build == "sweep":
        monkeypatch.setenv("XSDBA_HIP_LIN2D_BUILD", "sweep")
    if walk:
        monkeypatch.setenv("XSDBA_HIP_LIN2D_WALK", "1")
    return np.asarray(hip.qm_adjust_linear2d(*args, **kw))


def _agree(a, b, frac=1e-3, what="", atol=3e-6):
    nan_mis = np.isnan(a) != np.isnan(b)
    both = ~np.isnan(a) & ~np.isnan(b)
    bad = ~np.isclose(a[both], b[both], rtol=3e-6, atol=atol)
    assert nan_mis.sum() + bad.sum() <= frac * a.size, (what, int(nan_mis.sum()), int(bad.sum()), a.size)


def _queries(rng, hq, C, T, G):
    sim = (280 + 12 * rng.standard_normal((C, T))).astype(np.float32)
    newg = rng.uniform(0.5, G + 0.5, T)
    newg[:50] = np.round(newg[:50])                      # on the rows
    sim[:, 50:70] = hq[:, 3, 4][:, None]                 # a node's abscissa ... (NaN where the test removed the node)
    newg[50:60] = 4.0                                    # ... on its row (row 3 of the table is row 4 of the wrapped one)
    sim[:, 70:80] = 400.0                                # beyond the right ends
    sim[:, 80:90] = 100.0                                # beyond the left ends
    return sim, newg


@pytest.mark.parametrize("G,Q", [(12, 20), (4, 15), (12, 50)])
@pytest.mark.parametrize("extrap", ["constant", "nan"])
def test_small_tables_all_paths_agree_and_match_griddata(monkeypatch, G, Q, extrap):
    rng = np.random.default_rng(G * 100 + Q)
    C, T = 24, 3000
    af, hq = _table(rng, C, G, Q)
    af[1, 2, :3] = np.nan; hq[1, 2, :3] = np.nan         # a shorter row
    af[2, 5 % G] = np.nan; hq[2, 5 % G] = np.nan         # an empty row
    af[3, 7 % G, 1:] = np.nan; hq[3, 7 % G, 1:] = np.nan  # a row of one node: the sweep inside the new kernel
    hq[4, 1, 5] = hq[4, 1, 4]                            # a repeated abscissa: the first is kept
    sim, newg = _queries(rng, hq, C, T, G)
    gcoord = np.arange(0, G + 2.0)
    ref = _run(monkeypatch, "sweep", True, sim, af, hq, gcoord, newg, extrap, "+")
    for build, walk in (("rows", False), ("rows", True), ("sweep", False)):
        got = _run(monkeypatch, build, walk, sim, af, hq, gcoord, newg, extrap, "+")
        _agree(got, ref, what=f"build={build} walk={walk}")
    # and scipy itself on a few cells (scen - sim is the factor up to the float32 rounding of scen: half an ulp of ~300)
    got = _run(monkeypatch, "rows", False, sim, af, hq, gcoord, newg, "nan", "+")
    for c in (0, 1, 2, 5):
        tab_x = np.concatenate([hq[c][-1:], hq[c], hq[c][:1]]).astype(np.float64)
        tab_v = np.concatenate([af[c][-1:], af[c], af[c][:1]]).astype(np.float64)
        yy = np.repeat(gcoord[:, None], Q, 1)
        ok = ~np.isnan(tab_x) & ~np.isnan(tab_v)
        exp = griddata((tab_x[ok], yy[ok]), tab_v[ok], (sim[c].astype(np.float64), newg), method="linear")
        _agree(got[c] - sim[c], exp.astype(np.float32), frac=3e-3, what=f"griddata cell {c}", atol=1e-4)


def test_dayofyear_table_paths_agree(monkeypatch):
    rng = np.random.default_rng(5)
    C, T, G, Q = 6, 4000, 365, 20
    af, hq = _table(rng, C, G, Q)
    sim = (280 + 12 * rng.standard_normal((C, T))).astype(np.float32)
    newg = rng.uniform(0.5, G + 0.5, T)
    newg[:100] = np.round(newg[:100])
    gcoord = np.arange(0, G + 2.0)
    ref = _run(monkeypatch, "sweep", True, sim, af, hq, gcoord, newg, "constant", "+")
    for build, walk in (("rows", False), ("rows", True)):
        _agree(_run(monkeypatch, build, walk, sim, af, hq, gcoord, newg, "constant", "+"), ref, what=f"build={build} walk={walk}")


def test_cubic_and_tables_that_take_the_walk(monkeypatch):
    rng = np.random.default_rng(11)
    C, T, G, Q = 8, 2000, 12, 20
    af, hq = _table(rng, C, G, Q)
    sim, newg = _queries(rng, hq, C, T, G)
    gcoord = np.arange(0, G + 2.0)
    ref = _run(monkeypatch, "sweep", True, sim, af, hq, gcoord, newg, "constant", "+", cubic=True)
    _agree(_run(monkeypatch, "rows", False, sim, af, hq, gcoord, newg, "constant", "+", cubic=True), ref, what="cubic")
    # row coordinates float32 cannot hold exactly: no strip lists, same values
    g3 = gcoord / 3.0
    ref = _run(monkeypatch, "sweep", True, sim, af, hq, g3, newg / 3.0, "constant", "+")
    _agree(_run(monkeypatch, "rows", False, sim, af, hq, g3, newg / 3.0, "constant", "+"), ref, what="thirds")
    # a float64 table
    af64, hq64 = af.astype(np.float64), hq.astype(np.float64)
    ref = _run(monkeypatch, "sweep", True, sim.astype(np.float64), af64, hq64, gcoord, newg, "constant", "+")
    _agree(_run(monkeypatch, "rows", False, sim.astype(np.float64), af64, hq64, gcoord, newg, "constant", "+"), ref, what="float64")


@pytest.mark.parametrize("spread", [0.02, 0.3, 300.0])
def test_dense_and_sparse_rows_many_flip_rounds(monkeypatch, spread):
    """rows much denser than they are apart (nearly every row edge of the strip triangulation is flipped: more flips wanted in the
    first rounds than the LDS list of wanted flips holds, ~150 rounds) and much sparser (hardly a flip): the row construction
    against the sweep, and against griddata"""
    rng = np.random.default_rng(int(spread * 100) + 3)
    C, T, G, Q = 16, 2000, 12, 20
    hq = np.sort(280 + spread * rng.standard_normal((C, G, Q)), axis=2).astype(np.float32)
    af = rng.standard_normal((C, G, Q)).astype(np.float32)
    sim = (280 + 1.5 * spread * rng.standard_normal((C, T))).astype(np.float32)
    newg = rng.uniform(0.5, G + 0.5, T)
    gcoord = np.arange(0, G + 2.0)
    ref = _run(monkeypatch, "sweep", True, sim, af, hq, gcoord, newg, "nan", "+")
    got = _run(monkeypatch, "rows", False, sim, af, hq, gcoord, newg, "nan", "+")
    _agree(got, ref, frac=3e-3, what="rows vs sweep", atol=1e-4)
    for c in (0, 7):
        tab_x = np.concatenate([hq[c][-1:], hq[c], hq[c][:1]]).astype(np.float64)
        tab_v = np.concatenate([af[c][-1:], af[c], af[c][:1]]).astype(np.float64)
        keep = np.ones_like(tab_x, bool)
        keep[:, 1:] = tab_x[:, 1:] > tab_x[:, :-1]       # of exactly repeated abscissae the first is kept
        yy = np.repeat(gcoord[:, None], Q, 1)
        exp = griddata((tab_x[keep], yy[keep]), tab_v[keep], (sim[c].astype(np.float64), newg), method="linear")
        _agree(got[c] - sim[c], exp.astype(np.float32), frac=1e-2, what=f"griddata cell {c}", atol=2e-4)
