"""Pins the CPU oracle (oracle/xsdba_oracle.c) to the golden vectors generated from the reference's
own kernel source (tests/golden/make_golden.py) and to the reference tests' known answers.

Tolerances (stated per SURVEY §9.10):
  * float64 quantiles, 1-D interpolation, 2-D nearest-node selection, ranks: exact (same arithmetic).
  * float32 quantiles: 2 ulp (the golden vectors were produced by plain-numpy execution of the
    reference source, which keeps the gamma>=0.5 lerp branch in float32, whereas the oracle follows
    numba's type-based promotion to float64 for that branch, nbutils.py:104).
  * DQM quantities (group-mean normalised): the reference sums float32 in numpy's pairwise order, the
    oracle accumulates in float64 -> 1e-5 relative + 2e-6 * data-scale absolute.
"""
from __future__ import annotations

import numpy as np
import pytest

import oracle
from conftest import assert_close, assert_close_or_tie

F32_ULP2 = 2.5e-7


def _gidx_tables(d):
    return d["gidx"], int(d["G"]), int(d["W"])


def test_quantile_1d_against_reference_kernel(golden):
    d = golden("quantile_1d")
    for i in range(int(d["n"])):
        arr, q, exp = d[f"arr_{i}"], d[f"q_{i}"], d[f"out_{i}"]
        got = oracle.quantile_groups(arr[None, :], np.zeros(arr.size, np.int32), 1, 1, q)[0, 0]
        rtol = 0.0 if arr.dtype == np.float64 else F32_ULP2
        assert_close(got, exp, rtol=rtol, what=f"quantile case {i} L={arr.size} {arr.dtype}")


def test_quantile_edge_cases(golden):
    # reference tests/test_nbutils.py:23-34
    d = golden("quantile_1d")
    q = np.linspace(0.1, 0.99, 50)
    one = np.array([[1.0] + [np.nan] * 100])
    got = oracle.quantile_groups(one, np.zeros(101, np.int32), 1, 1, q)[0, 0]
    np.testing.assert_array_equal(got, np.full_like(q, 1))
    np.testing.assert_array_equal(got, d["edge_one_out"])
    got = oracle.quantile_groups(np.full((1, 100), np.nan), np.zeros(100, np.int32), 1, 1, q)[0, 0]
    assert np.isnan(got).all() and np.isnan(d["edge_nan_out"]).all()


def test_quantile_matches_numpy_nanquantile():
    # reference tests/test_nbutils.py:12-21 (nbu.quantile == xarray/np nanquantile to 6 decimals)
    rng = np.random.default_rng(0)
    x = rng.gamma(2.0, 3.0, (4, 2190))
    x[rng.random(x.shape) < 0.05] = np.nan
    q = np.linspace(0.1, 0.99, 50)
    got = oracle.quantile_groups(x, np.zeros(2190, np.int32), 1, 1, q)[:, 0]
    np.testing.assert_array_almost_equal(got, np.nanquantile(x, q, axis=1).T, 6)


def test_interp_1d_against_reference(golden):
    d = golden("interp_1d")
    for i in range(int(d["n"])):
        newx, xq, yq = d[f"newx_{i}"], d[f"xq_{i}"], d[f"yq_{i}"]
        method, extrap = str(d[f"method_{i}"]), str(d[f"extrap_{i}"])
        got = oracle.qm_adjust(newx[None], yq[None, None], xq[None, None], None, method, extrap, kind=None)[0]
        assert_close(got, d[f"out_{i}"], rtol=0.0, what=f"interp1d case {i} {method}/{extrap}")


def test_interp_1d_known_answers():
    # reference tests/test_utils.py:68-113: expected 4.4 (constant extrapolation), 2.9 nearest, 2.95 linear
    xq = np.linspace(205, 229, num=25)
    yq = np.linspace(2, 4.4, num=25)
    newx = np.linspace(240, 200, num=41) - 0.5
    newx = np.where(newx > 201, newx, np.nan)
    for interp, expi in (("nearest", 2.9), ("linear", 2.95)):
        for extrap, expe in (("constant", 4.4), ("nan", np.nan)):
            out = oracle.qm_adjust(newx[None], yq[None, None], xq[None, None], None, interp, extrap, kind=None)[0]
            if np.isnan(expe):
                assert np.isnan(out[0])
            else:
                assert out[0] == expe
            np.testing.assert_allclose(out[25], expi)
            assert np.isnan(out[-1])


def test_interp_2d_nearest_against_griddata(golden):
    d = golden("interp_2d_nearest")
    for i in range(int(d["n"])):
        sim, hq, af = d[f"sim_{i}"], d[f"hq_{i}"], d[f"af_{i}"]
        gidx = (d[f"newg_{i}"] - 1).astype(np.int32)
        got = oracle.qm_adjust(sim, af, hq, gidx, "nearest", str(d[f"extrap_{i}"]), kind=None)
        exp = d[f"out_{i}"]
        # exact distance ties are implementation-defined in cKDTree: none allowed to differ
        # by more than a handful of points, and never in NaN pattern
        assert_close_or_tie(got, exp, sim, hq, af, gidx, None, 0.0, what=f"interp2d case {i}")


def test_rank_against_reference(golden):
    d = golden("rank")
    for i in range(int(d["n"])):
        arr = d[f"arr_{i}"]
        got = oracle.rank_groups(arr, np.zeros(arr.shape[1], np.int32), 1, 1, False)
        assert_close(got, d[f"out_{i}"], rtol=0.0, atol=0.0, what=f"rank case {i}")


PIPES = ["time_add", "time_mul", "month_add", "month_mul", "doy31_add", "doy31_mul", "doy1_add", "doy1_mul",
         "month5_add", "month5_mul"]


@pytest.mark.parametrize("name", PIPES)
def test_pipeline_train(golden, name):
    d = golden(f"pipeline_{name}")
    gidx, G, W = _gidx_tables(d)
    kind = str(d["kind"])
    af, hq = oracle.eqm_train(d["ref"], d["hist"], gidx, G, W, d["q"], kind)
    assert_close(hq, d["eqm_hist_q"], rtol=F32_ULP2, what="eqm hist_q")
    # additive af = ref_q - hist_q: each operand carries the <=2 ulp float32 slack of the lerp branch
    scale_q = float(np.nanmax(np.abs(d["eqm_hist_q"])))
    assert_close(af, d["eqm_af"], rtol=2e-6, atol=4 * 1.2e-7 * scale_q if kind == "+" else 0, what="eqm af")
    if "dqm_af" in d.files:
        scale = float(np.nanmax(np.abs(d["ref"])))
        af, hq, sc = oracle.dqm_train(d["ref"], d["hist"], gidx, G, W, d["q"], kind)
        atol = 2e-6 * scale if kind == "+" else 2e-6
        assert_close(sc, d["dqm_scaling"], rtol=1e-5, atol=atol, what="dqm scaling")
        assert_close(hq, d["dqm_hist_q"], rtol=1e-5, atol=atol, what="dqm hist_q")
        assert_close(af, d["dqm_af"], rtol=1e-5, atol=2 * atol, what="dqm af")


@pytest.mark.parametrize("name", PIPES)
def test_pipeline_eqm_adjust(golden, name):
    d = golden(f"pipeline_{name}")
    gidx, G, W = _gidx_tables(d)
    kind = str(d["kind"])
    for key in d.files:
        if not key.startswith("eqm_scen_"):
            continue
        _, _, interp, extrap = key.split("_")
        got = oracle.qm_adjust(d["sim"], d["eqm_af"], d["eqm_hist_q"], gidx, interp, extrap, kind)
        if G > 1:
            assert_close_or_tie(got, d[key], d["sim"], d["eqm_hist_q"], d["eqm_af"], gidx, kind, 1e-6, what=key)
        else:
            assert_close(got, d[key], rtol=1e-6, what=key)


@pytest.mark.parametrize("name", [p for p in PIPES if not p.startswith("month5")])
def test_pipeline_qdm_adjust(golden, name):
    d = golden(f"pipeline_{name}")
    gidx, G, W = _gidx_tables(d)
    kind = str(d["kind"])
    for rw in (0, 1):
        for interp in ("nearest", "linear"):
            key = f"qdm_scen_{interp}_rw{rw}"
            if key not in d.files:
                continue
            scen, sim_q = oracle.qdm_adjust(d["sim"], d["eqm_af"], d["q"], gidx, W, bool(rw), interp, "constant", kind)
            assert_close(sim_q, d[f"qdm_simq_rw{rw}"], rtol=0.0, atol=0.0, what=f"sim_q rw{rw}")
            if G > 1:  # sim_q = k/(N-1) can sit exactly midway between two float32 nodes: cKDTree tie
                assert_close_or_tie(scen, d[key], d["sim"], d["q"], d["eqm_af"], gidx, kind, 1e-6, what=key,
                                    newx=sim_q, max_frac=5e-3)
            else:
                assert_close(scen, d[key], rtol=1e-6, what=key)


@pytest.mark.parametrize("name", [p for p in PIPES if "dqm" and not p.startswith("month5")])
def test_pipeline_dqm_adjust(golden, name):
    d = golden(f"pipeline_{name}")
    if "dqm_af" not in d.files:
        pytest.skip("no DQM vectors for this grouping")
    gidx, G, W = _gidx_tables(d)
    kind = str(d["kind"])
    scale = float(np.nanmax(np.abs(d["sim"])))
    for interp in ("nearest", "linear"):
        key = f"dqm_scen_{interp}"
        if key not in d.files:
            continue
        scen, trend = oracle.dqm_adjust(d["sim"], d["dqm_af"], d["dqm_hist_q"], d["dqm_scaling"], gidx, W, d["tns"],
                                        1, interp, "constant", kind)
        assert_close(trend, d["dqm_trend"], rtol=1e-5, atol=2e-6 * scale, what="trend")
        # a detrended value sitting within rounding distance of a table node may select the neighbouring
        # node (documented knife-edge): allow a vanishing fraction of such points, hold the rest to 1e-5
        exp = d[key]
        assert np.array_equal(np.isnan(scen), np.isnan(exp))
        ok = ~np.isnan(exp)
        err = np.abs(scen[ok].astype(np.float64) - exp[ok]) / np.maximum(np.abs(exp[ok]), 1e-30)
        assert (err > 1e-5).mean() < 2e-3, f"{key}: {(err > 1e-5).mean():.2e} of points beyond 1e-5"


def test_known_answer_qm(golden):
    # reference tests/test_adjustment.py:947-989 (TestQM.test_quantiles): af to 1 decimal away from the ends
    from scipy.stats import norm, uniform

    d = golden("known_answer_qm")
    x, y, q = d["x"], d["y"], d["q50"]
    xd, yd = uniform(loc=10, scale=1), norm(loc=12, scale=1)
    gidx = np.zeros(x.size, np.int32)
    for kind, tag in (("+", "add"), ("*", "mul")):
        af, hq = oracle.eqm_train(y[None], x[None], gidx, 1, 1, q, kind)
        expected = (yd.ppf(q) - xd.ppf(q)) if kind == "+" else (yd.ppf(q) / xd.ppf(q))
        np.testing.assert_array_almost_equal(af[0, 0, 2:-2], expected[2:-2], 1)
        assert_close(af, d[f"eqm_af_{tag}"], rtol=0.0, what="af vs reference kernel (float64)")
        p = oracle.qm_adjust(x[None], af, hq, None, "linear", "constant", kind)
        assert_close(p, d[f"eqm_p_linear_{tag}"], rtol=1e-15, what="p vs reference kernel")
        middle = (d["u"] > 1e-2) & (d["u"] < 0.99)
        np.testing.assert_array_almost_equal(p[0][middle], y[middle], 1)


def test_vecquantiles_matches_numpy():
    # a5 (nbutils.py:151-161): numba's np.nanquantile == numpy's to float64 rounding (numba itself cannot run here)
    import warnings

    rng = np.random.default_rng(2)
    x = rng.gamma(2, 3, (60, 777))
    x[rng.random(x.shape) < 0.1] = np.nan
    x[3] = np.nan
    x[4, 1:] = np.nan
    q = rng.random(60)
    q[7], q[8], q[9] = np.nan, 0.0, 1.0
    got = oracle.vecquantiles(x, q)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        exp = np.array([np.nanquantile(x[i], q[i]) if not np.isnan(q[i]) else np.nan for i in range(60)])
    assert_close(got, exp, rtol=1e-13, what="vecquantiles vs numpy")


# ------------------------------------------------------------------------------------------------
# f1 pre-processing rows: the reference draws from an unseeded generator, so there are no golden vectors to pin; the
# oracle is held to the assertions of the reference's own tests instead (tests/test_processing.py:31-64, 91-132)
# ------------------------------------------------------------------------------------------------
def test_oracle_jitter_reference_assertions():
    da = np.array([[0.5, 2.1, np.nan]])
    out = oracle.jitter(da, lower=1.0, upper=2.0, maximum=3.0, seed=1)
    assert da[0, 0] != out[0, 0] and 0 < out[0, 0] < 1
    assert da[0, 1] != out[0, 1] and 2 <= out[0, 1] < 3 and np.isnan(out[0, 2])
    for dtype, delta in ((np.float64, 1e-8), (np.float32, 1e-6)):  # narrow intervals, :66-75
        for test_val in (1e-6, 1, 100, 1e6):
            x = (test_val + np.zeros((1, 1000))).astype(dtype)
            hi = oracle.jitter(x, upper=test_val * (1 - delta), maximum=test_val, seed=2)
            lo = oracle.jitter(x, lower=test_val * (1 + delta), minimum=test_val, seed=3)
            assert (hi < test_val).all() and (lo > test_val).all()


def test_oracle_adapt_freq_reference_assertions():
    rng = np.random.default_rng(0)
    n = 31 * 365 + 8
    pr = rng.integers(0, 100, size=(3, n)).astype(np.float64)
    prsim, prref = np.where(pr < 20, pr / 20, pr), np.where(pr < 10, pr / 20, pr)
    doy = np.arange(n) % 365
    cum = np.concatenate([[0], np.cumsum([31, 28, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31])])
    month = (np.searchsorted(cum, doy, side="right") - 1).astype(np.int32)
    sim_ad, pth, dP0, P0_ref, P0_hist = oracle.adapt_freq(prsim, prref, month, 12, 1, 1.0, seed=7)
    zeros = prsim <= 1
    np.testing.assert_allclose((sim_ad[zeros] > 1).sum() / (n * 3 * 0.2), 0.5, atol=0.1)
    corrected = sim_ad[zeros & (sim_ad > 1)]
    assert ((corrected < 20.1) & (corrected > 1)).all()
    np.testing.assert_array_equal(sim_ad[prsim > 20.5], prsim[prsim > 20.5])
    np.testing.assert_allclose(pth, 20, rtol=0.05)
    np.testing.assert_allclose(dP0, 0.5, atol=0.25)
    # adjust branch with the trained tables: same decisions on the same data and seed
    again = oracle.adapt_freq(prsim, None, month, 12, 1, 1.0, seed=7, P0_ref=P0_ref, P0_hist=P0_hist, pth=pth)
    np.testing.assert_array_equal(again[0], sim_ad)


# ------------------------------------------------------------------------------------------------
# round 2: rows that had no reference-generated vector (Scaling, LOCI, max_tail_factor, _vecquantiles, fractional group
# coordinates, the window / group bookkeeping itself)
# ------------------------------------------------------------------------------------------------
def _lin_nodes(i, G):
    """(g_lo, g_hi, frac) of utils.broadcast's linear interpolation between month centres, as xsdba_amd derives them"""
    hi = np.ceil(i)
    lo = hi - 1
    return np.mod(lo - 1, G).astype(np.int32), np.mod(hi - 1, G).astype(np.int32), i - lo


@pytest.mark.parametrize("gname", ["month", "doy31", "time"])
@pytest.mark.parametrize("tag", ["add", "mul"])
def test_scaling_against_reference_composition(golden, gname, tag):
    d = golden("families_scaling_loci_tail")
    kind = "+" if tag == "add" else "*"
    k = f"{gname}_{tag}"
    gidx, (G, W) = d[f"{gname}_gidx"], d[f"{gname}_GW"]
    ref, hist, sim = d[f"{k}_ref"], d[f"{k}_hist"], d[f"{k}_sim"]
    mh, mr = oracle.group_means(hist, gidx, int(G), int(W)), oracle.group_means(ref, gidx, int(G), int(W))
    af = (mr - mh) if kind == "+" else (mr / mh)
    scale = float(np.nanmax(np.abs(ref)))
    # the reference sums float32 in numpy's pairwise order, the oracle in float64: 1e-5 relative (+ data-scale absolute for '+')
    assert_close(af, d[f"{k}_scaling_af"], rtol=1e-5, atol=2e-6 * scale if kind == "+" else 0.0, what="scaling af")
    got = oracle.scaling_adjust(sim, d[f"{k}_scaling_af"], gidx, kind)
    assert_close(got, d[f"{k}_scaling_scen_nearest"], rtol=1e-6, what="scaling scen nearest")
    if gname == "month":
        g_lo, g_hi, frac = _lin_nodes(d["month_index_interp"], 12)
        got = oracle.scaling_adjust(sim, d[f"{k}_scaling_af"], g_lo, kind, g_hi=g_hi, frac=frac)
        assert_close(got, d[f"{k}_scaling_scen_linear"], rtol=1e-6, what="scaling scen linear")


@pytest.mark.parametrize("gname", ["month", "doy31", "time"])
def test_loci_against_reference_composition(golden, gname):
    d = golden("families_scaling_loci_tail")
    k = f"{gname}_mul"
    gidx, (G, W) = d[f"{gname}_gidx"], d[f"{gname}_GW"]
    ref, hist, sim, thresh = d[f"{k}_ref"], d[f"{k}_hist"], d[f"{k}_sim"], float(d[f"{k}_loci_thresh"])
    af, sth = oracle.loci_train(ref, hist, gidx, int(G), int(W), thresh)
    assert_close(sth, d[f"{k}_loci_hist_thresh"], rtol=2e-6, what="loci hist_thresh")
    assert_close(af, d[f"{k}_loci_af"], rtol=2e-5, what="loci af")
    got = oracle.loci_adjust(sim, d[f"{k}_loci_af"], d[f"{k}_loci_hist_thresh"], gidx, thresh)
    assert_close(got, d[f"{k}_loci_scen_nearest"], rtol=1e-6, atol=1e-6, what="loci scen nearest")
    if gname == "month":
        g_lo, g_hi, frac = _lin_nodes(d["month_index_interp"], 12)
        got = oracle.loci_adjust(sim, d[f"{k}_loci_af"], d[f"{k}_loci_hist_thresh"], g_lo, thresh, g_hi=g_hi, frac=frac)
        assert_close(got, d[f"{k}_loci_scen_linear"], rtol=2e-6, atol=2e-6, what="loci scen linear")


@pytest.mark.parametrize("gname", ["month", "doy31"])
@pytest.mark.parametrize("tag", ["add", "mul"])
def test_tail_mask_against_reference_composition(golden, gname, tag):
    d = golden("families_scaling_loci_tail")
    k = f"{gname}_{tag}"
    gidx = d[f"{gname}_gidx"]
    for f in (1.0, 1.02):
        got = oracle.tail_mask(d[f"{k}_sim"], d[f"{k}_eqm_scen"], d[f"{k}_eqm_hist_q"], gidx, f)
        exp = d[f"{k}_tailmask_{f}"]
        assert_close(got, exp, rtol=0.0, what=f"tail mask {f}")
        if f == 1.0:
            assert (got != d[f"{k}_eqm_scen"])[~np.isnan(got)].any()  # the mask does bite


def test_vecquantiles_against_reference_kernel(golden):
    d = golden("vecquantiles")
    for i in range(int(d["n"])):
        arr, rnk, exp = d[f"arr_{i}"], d[f"rnk_{i}"], d[f"out_{i}"]
        got = oracle.vecquantiles(arr, rnk)
        # numpy's nanquantile lerps in the array dtype, numba's (what the reference runs, restated by the oracle) in float64
        assert_close(got, exp, rtol=0.0 if arr.dtype == np.float64 else F32_ULP2, atol=1e-12, what=f"vecquantiles {i}")


def test_interp_2d_fractional_group_coordinate(golden):
    d = golden("interp_2d_fractional")
    for i in range(int(d["n"])):
        sim, hq, af, newg = d[f"sim_{i}"], d[f"hq_{i}"], d[f"af_{i}"], d[f"newg_{i}"]
        method, extrap = str(d[f"method_{i}"]), str(d[f"extrap_{i}"])
        G = hq.shape[1]
        oldg = np.repeat(np.arange(0, G + 2, dtype=np.float64)[:, None], hq.shape[2], axis=1)
        for c in range(sim.shape[0]):
            ext = lambda t: np.concatenate([t[c, -1:], t[c], t[c, :1]])  # noqa: E731
            got = oracle.interp_on_quantiles_2d(sim[c], newg, ext(hq), ext(af), oldg, method, extrap)
            assert_close(got, d[f"out_{i}"][c], rtol=1e-6, atol=1e-7, what=f"case {i} {method}/{extrap} cell {c}")


def test_window_group_bookkeeping_hand_case(golden):
    """tests/test_base.py:76-120 (Grouper.apply("mean") with and without window=5 on arange(366)): expectations computed
    with explicit loops over real dates in make_golden.py; January without a window is mean(0..30) = 15"""
    d = golden("window_groups_handcase")
    x = d["x"][None, :]
    assert d["plain"][0] == 15.0
    np.testing.assert_allclose(oracle.group_means(x, d["gidx"], 12, 1)[0], d["plain"], rtol=1e-14)
    np.testing.assert_allclose(oracle.group_means(x, d["gidx"], 12, 5)[0], d["win5"], rtol=1e-14)
    np.testing.assert_allclose(oracle.group_means(x, d["doy_gidx"], 366, 5)[0], d["doy_win5"], rtol=1e-14)
    # the quantile path gathers the same rows: the median of a windowed day-of-year row of arange is its centre value
    med = oracle.quantile_groups(x, d["doy_gidx"], 366, 5, np.array([0.5]))[0, :, 0]
    np.testing.assert_allclose(med[2:-2], d["x"][2:-2], rtol=1e-14)


# ---- round 3: the deterministic half of the pre-processing steps, from the reference's own kernels (make_golden.py round3)
def test_adapt_freq_deterministic_half_against_reference_kernels(golden):
    from preproc_golden_checks import check_adapt_freq

    check_adapt_freq(oracle, golden("preproc_deterministic"))


def test_jitter_bounds_and_masks_against_reference_lines(golden):
    from preproc_golden_checks import check_jitter

    check_jitter(oracle, golden("preproc_deterministic"))


def test_window_group_bookkeeping_hand_case_nonreducing(golden):
    """rank and trend through the non-reducing path of Grouper.apply (centre slot, time order): explicit-loop expectations"""
    d = golden("window_groups_handcase_nonreducing")
    x, W = d["x"][None, :], int(d["W"])
    r = oracle.rank_groups(x, d["gidx"], 12, W, rank_window=True)
    assert np.array_equal(np.isnan(r[0]), np.isnan(d["rank_pct"]))
    np.testing.assert_allclose(r[0], d["rank_pct"], rtol=1e-14, atol=1e-15)
    tr = oracle.polydetrend_trend(x, d["gidx"], 12, W, d["time_ns"], 1)
    np.testing.assert_allclose(tr[0], d["trend"], rtol=1e-12)


# ---- round 4: hand cases for the groupings that had none (make_golden.py round4: python datetime + explicit loops)
def test_week_and_season_bookkeeping_hand_case(golden):
    """`time.season` / `time.week` group indexes of the host mirror (base.py:196-198, 312-327 of the reference) against
    python's own datetime, and the oracle's row gathering on them: seasonal / weekly means with and without a window."""
    import xsdba_amd as xs
    from xsdba_amd.labeled import DataArray

    d = golden("week_season_handcase")
    x = d["x"]
    da = DataArray(x, dims=("time",), coords={"time": xs.date_range("2000-01-01", x.size, "standard").values})
    gs, G = xs.Grouper("time.season").group_index(da)
    assert G == 4
    np.testing.assert_array_equal(gs, d["season_gidx"])
    gw, G = xs.Grouper("time.week").group_index(da)
    assert G == 53
    np.testing.assert_array_equal(gw, d["week_gidx"])
    x2 = x[None, :]
    np.testing.assert_allclose(oracle.group_means(x2, gs, 4, 1)[0], d["season_plain"], rtol=1e-14)
    np.testing.assert_allclose(oracle.group_means(x2, gs, 4, 5)[0], d["season_win5"], rtol=1e-14)
    np.testing.assert_allclose(oracle.group_means(x2, gw, 53, 1)[0], d["week_plain"], rtol=1e-14, equal_nan=True)
    np.testing.assert_allclose(oracle.group_means(x2, gw, 53, 3)[0], d["week_win3"], rtol=1e-14, equal_nan=True)
    med = oracle.quantile_groups(x2, gs, 4, 1, np.array([0.5]))[0, :, 0]
    np.testing.assert_allclose(med, d["season_median"], rtol=1e-14)
