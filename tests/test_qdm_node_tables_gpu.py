"""Pass D of k_qdm_cell with the node tables made once per launch (k_qdm_node_tables: the map 2 * rank -> node of a ranking row
of N valid values with untied extremes, one table per distinct row length of the plan) against the oracle (reference
_adjustment.py:783-886, utils.py:573-638 restated) and against the same kernel with every cell building its own table
(XSDBA_HIP_QDM_TABLES=0), bit for bit.  Cells of every kind in one launch: all groups on the launch's tables; ties at the extremes
of some / all groups (zero-inflated, capped); NaNs in some groups (their N differs from the row length); the middle group on the
launch's table while others are not; groups without a valid value; an all-NaN cell; month (three row lengths), season,
day-of-year windows with rank_window (one row length, the two-stream walk)."""
import os

import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu


def _doy_month(T):
    d0 = np.arange(T) % 365
    cum = np.concatenate([[0], np.cumsum([31, 28, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31])])
    return d0.astype(np.int32), (np.searchsorted(cum, d0, side="right") - 1).astype(np.int32)


def _grouping(name, T):
    doy, mon = _doy_month(T)
    if name == "month":
        return mon, 12, 1, False
    if name == "season":
        return ((mon + 1) % 12 // 3).astype(np.int32), 4, 1, False
    if name == "doy31_rw":
        return doy, 365, 31, True
    raise KeyError(name)


def _cells(T, gidx, G, rng, dtype):
    """one cell per situation"""
    C = 9
    sim = rng.gamma(2.0, 2.0, (C, T)).astype(dtype)
    sim[1, rng.random(T) < 0.5] = 0.0                      # ties at every group's minimum
    sim[2] = np.minimum(sim[2], dtype(4.0))                # ... maximum
    sim[3, rng.random(T) < 0.03] = np.nan                  # NaNs nearly everywhere: N differs from the row lengths
    sim[4, (gidx == 0) & (rng.random(T) < 0.2)] = np.nan   # NaNs in the first group only: the others stay on the launch's tables
    m = gidx == G // 2
    sim[5, m] = np.where(rng.random(m.sum()) < 0.5, dtype(1.0), sim[5, m])   # ties inside the middle group only (not at its extremes, mostly)
    sim[6, gidx == G // 2] = np.nan                        # the middle group without a value
    sim[7] = np.nan                                        # no value at all
    sim[8, gidx != 1] = np.minimum(sim[8, gidx != 1], dtype(3.0))   # every group but one tied at its maximum: the middle group's table is the cell's
    return sim


def _run(hip, tables, *args, **kw):
    """tables: True (the launch's tables, copied into the LDS where they fit), "global" (read in place), False (none)"""
    if tables is True:
        os.environ.pop("XSDBA_HIP_QDM_TABLES", None)
    else:
        os.environ["XSDBA_HIP_QDM_TABLES"] = "global" if tables == "global" else "0"
    try:
        scen, _ = hip.qdm_adjust(*args, want_sim_q=False, **kw)
    finally:
        os.environ.pop("XSDBA_HIP_QDM_TABLES", None)
    return np.asarray(scen)


@pytest.mark.parametrize("T", [1825, 3000, 7300, 9000, 10950])   # (whole years and not: 3 .. 12 distinct row lengths)
@pytest.mark.parametrize("gname", ["month", "season", "doy31_rw"])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_launch_tables_vs_cell_tables_and_oracle(T, gname, dtype):
    from xsdba_amd import hip

    rng = np.random.default_rng(T + len(gname))
    gidx, G, W, rw = _grouping(gname, T)
    sim = _cells(T, gidx, G, rng, dtype)
    C = sim.shape[0]
    for Q, kind, extrap in ((20, "*", "constant"), (15, "+", "nan")):
        q = ((np.arange(Q) + 0.5) / Q).astype(dtype)
        af = (1.0 + 0.3 * rng.standard_normal((C, G, Q))).astype(dtype)
        a = _run(hip, True, sim, af, q, gidx, W, rw, "nearest", extrap, kind)
        b = _run(hip, False, sim, af, q, gidx, W, rw, "nearest", extrap, kind)
        g = _run(hip, "global", sim, af, q, gidx, W, rw, "nearest", extrap, kind)
        exp, _ = oracle.qdm_adjust(sim, af, q, gidx, W, rw, "nearest", extrap, kind)
        np.testing.assert_array_equal(a, exp)
        np.testing.assert_array_equal(b, exp)
        np.testing.assert_array_equal(g, exp)
        assert a.dtype == dtype


def test_uneven_quantile_nodes_and_many_row_lengths():
    """nodes that are not equally spaced (the table must be made of THESE nodes); 40 groups of 17 distinct sizes (16 classes at
    most: the groups beyond them go the cell's way)"""
    from xsdba_amd import hip

    rng = np.random.default_rng(77)
    sizes = np.concatenate([np.arange(60, 77), rng.integers(60, 77, 23)])
    gidx = rng.permutation(np.repeat(np.arange(40), sizes)).astype(np.int32)
    T, G, C = len(gidx), 40, 4
    sim = rng.gamma(2.0, 2.0, (C, T)).astype(np.float32)
    sim[1, rng.random(T) < 0.3] = 0.0
    q = np.sort(rng.random(24)).astype(np.float32)
    q = q * np.float32(0.9) + np.float32(0.05)
    af = (1.0 + 0.3 * rng.standard_normal((C, G, 24))).astype(np.float32)
    a = _run(hip, True, sim, af, q, gidx, 1, False, "nearest", "constant", "*")
    b = _run(hip, False, sim, af, q, gidx, 1, False, "nearest", "constant", "*")
    exp, _ = oracle.qdm_adjust(sim, af, q, gidx, 1, False, "nearest", "constant", "*")
    np.testing.assert_array_equal(a, exp)
    np.testing.assert_array_equal(b, exp)


def _cells_one_group(T, rng, dtype):
    C = 8
    sim = rng.gamma(2.0, 2.0, (C, T)).astype(dtype)
    sim[1, rng.random(T) < 0.5] = 0.0                        # ties at the minimum: the cell's own evaluation
    sim[2] = np.minimum(sim[2], dtype(4.0))                  # ... at the maximum
    sim[3, rng.random(T) < 0.03] = np.nan                    # NaNs: N differs from the launch's
    sim[4] = np.round(sim[4], 1)                             # ties inside (the extremes untied, mostly): the launch's table
    sim[5] = np.nan                                          # no value at all
    sim[6, 5:] = np.nan                                      # five values
    return sim                                               # (7: NaN among its factors, set by the caller)


@pytest.mark.parametrize("T", [365, 1825, 2001, 4097, 7300, 10950, 11264])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_single_group_table_vs_elements_and_oracle(T, dtype):
    """group="time", interp="nearest": pass D by the launch's one table (utils._interp_on_quantiles_1D's nearest: bounds halfway
    between the nodes, a percent rank ON a bound goes down — T = 2001 with 20 even nodes puts 19 percent ranks exactly there)
    against the element-by-element evaluation (XSDBA_HIP_QDM_TABLES=0) and the oracle, bit for bit"""
    from xsdba_amd import hip

    rng = np.random.default_rng(T)
    gidx = np.zeros(T, np.int32)
    sim = _cells_one_group(T, rng, dtype)
    C = sim.shape[0]
    for Q, kind, extrap, even in ((20, "*", "constant", True), (15, "+", "nan", True), (33, "*", "nan", False), (1, "+", "constant", True)):
        q = ((np.arange(Q) + 0.5) / Q).astype(dtype) if even else (np.sort(rng.random(Q)) * 0.9 + 0.05).astype(dtype)
        af = (1.0 + 0.3 * rng.standard_normal((C, 1, Q))).astype(dtype)
        af[7, 0, Q // 2] = np.nan
        a = _run(hip, True, sim, af, q, gidx, 1, False, "nearest", extrap, kind)
        b = _run(hip, False, sim, af, q, gidx, 1, False, "nearest", extrap, kind)
        exp, _ = oracle.qdm_adjust(sim, af, q, gidx, 1, False, "nearest", extrap, kind)
        np.testing.assert_array_equal(a, exp)
        np.testing.assert_array_equal(b, exp)
        assert a.dtype == dtype
    # the other interpolations of one group do not take the table
    q = ((np.arange(20) + 0.5) / 20).astype(dtype)
    af = (1.0 + 0.3 * rng.standard_normal((C, 1, 20))).astype(dtype)
    a = _run(hip, True, sim, af, q, gidx, 1, False, "linear", "constant", "*")
    exp, _ = oracle.qdm_adjust(sim, af, q, gidx, 1, False, "linear", "constant", "*")
    np.testing.assert_allclose(a, exp, rtol=1e-6 if dtype == np.float32 else 1e-13, equal_nan=True)
