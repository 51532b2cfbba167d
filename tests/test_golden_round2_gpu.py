"""GPU parity (through the C ABI) against the reference-generated golden vectors added in round 2
(tests/golden/make_golden.py round2): Scaling and LOCI (_adjustment.py:889-974), the max_tail_factor mask (:648-658,
672-673), nbutils._vecquantiles (nbutils.py:151-161), grouped linear / cubic interpolation at the FRACTIONAL group
coordinate of Grouper.get_index(interp=True) (base.py:308-320, utils.py:380-400), and the window / group bookkeeping on
the reference's own arange(366) case (tests/test_base.py:76-120).  The same vectors pin the oracle on the CPU side
(tests/test_oracle_golden.py)."""
from __future__ import annotations

import numpy as np
import pytest

from conftest import assert_close

pytestmark = pytest.mark.gpu
F32_ULP2 = 2.5e-7


@pytest.fixture(scope="module")
def hip():
    from xsdba_amd import hip as h

    h.default_context(0)
    return h


def _lin_nodes(i, G):
    hi = np.ceil(i)
    lo = hi - 1
    return np.mod(lo - 1, G).astype(np.int32), np.mod(hi - 1, G).astype(np.int32), i - lo


@pytest.mark.parametrize("gname", ["month", "doy31", "time"])
@pytest.mark.parametrize("tag", ["add", "mul"])
def test_scaling(hip, golden, gname, tag):
    d = golden("families_scaling_loci_tail")
    kind = "+" if tag == "add" else "*"
    k = f"{gname}_{tag}"
    gidx, (G, W) = d[f"{gname}_gidx"], d[f"{gname}_GW"]
    ref, hist, sim = d[f"{k}_ref"], d[f"{k}_hist"], d[f"{k}_sim"]
    mh, mr = hip.group_means(hist, gidx, int(G), int(W)), hip.group_means(ref, gidx, int(G), int(W))
    af = (mr - mh) if kind == "+" else (mr / mh)
    scale = float(np.nanmax(np.abs(ref)))
    assert_close(af, d[f"{k}_scaling_af"], rtol=1e-5, atol=2e-6 * scale if kind == "+" else 0.0, what="scaling af")
    got = hip.scaling_adjust(sim, d[f"{k}_scaling_af"], gidx if G > 1 else None, kind)
    assert_close(got, d[f"{k}_scaling_scen_nearest"], rtol=1e-6, what="scaling scen nearest")
    if gname == "month":
        g_lo, g_hi, frac = _lin_nodes(d["month_index_interp"], 12)
        got = hip.scaling_adjust(sim, d[f"{k}_scaling_af"], g_lo, kind, g_hi=g_hi, frac=frac)
        assert_close(got, d[f"{k}_scaling_scen_linear"], rtol=1e-6, what="scaling scen linear")


@pytest.mark.parametrize("gname", ["month", "doy31", "time"])
def test_loci(hip, golden, gname):
    d = golden("families_scaling_loci_tail")
    k = f"{gname}_mul"
    gidx, (G, W) = d[f"{gname}_gidx"], d[f"{gname}_GW"]
    ref, hist, sim, thresh = d[f"{k}_ref"], d[f"{k}_hist"], d[f"{k}_sim"], float(d[f"{k}_loci_thresh"])
    af, sth = hip.loci_train(ref, hist, gidx, int(G), int(W), thresh)
    assert_close(sth, d[f"{k}_loci_hist_thresh"], rtol=2e-6, what="loci hist_thresh")
    assert_close(af, d[f"{k}_loci_af"], rtol=2e-5, what="loci af")
    got = hip.loci_adjust(sim, d[f"{k}_loci_af"], d[f"{k}_loci_hist_thresh"], gidx if G > 1 else None, thresh)
    assert_close(got, d[f"{k}_loci_scen_nearest"], rtol=1e-6, atol=1e-6, what="loci scen nearest")
    if gname == "month":
        g_lo, g_hi, frac = _lin_nodes(d["month_index_interp"], 12)
        got = hip.loci_adjust(sim, d[f"{k}_loci_af"], d[f"{k}_loci_hist_thresh"], g_lo, thresh, g_hi=g_hi, frac=frac)
        assert_close(got, d[f"{k}_loci_scen_linear"], rtol=2e-6, atol=2e-6, what="loci scen linear")


@pytest.mark.parametrize("gname", ["month", "doy31"])
@pytest.mark.parametrize("tag", ["add", "mul"])
def test_tail_mask(hip, golden, gname, tag):
    d = golden("families_scaling_loci_tail")
    k = f"{gname}_{tag}"
    for f in (1.0, 1.02):
        got = hip.tail_mask(d[f"{k}_sim"], d[f"{k}_eqm_scen"], d[f"{k}_eqm_hist_q"], d[f"{gname}_gidx"], f)
        assert_close(got, d[f"{k}_tailmask_{f}"], rtol=0.0, what=f"tail mask {f}")


def test_vecquantiles(hip, golden):
    d = golden("vecquantiles")
    for i in range(int(d["n"])):
        arr, rnk, exp = d[f"arr_{i}"], d[f"rnk_{i}"], d[f"out_{i}"]
        got = hip.vecquantiles(arr, rnk)
        assert_close(got, exp, rtol=0.0 if arr.dtype == np.float64 else F32_ULP2, atol=1e-12, what=f"vecquantiles {i}")


@pytest.mark.parametrize("method", ["linear", "cubic"])
def test_grouped_interp_at_fractional_group_coordinate(hip, golden, method):
    d = golden("interp_2d_fractional")
    seen = 0
    for i in range(int(d["n"])):
        if str(d[f"method_{i}"]) != method:
            continue
        sim, hq, af, newg, extrap = d[f"sim_{i}"], d[f"hq_{i}"], d[f"af_{i}"], d[f"newg_{i}"], str(d[f"extrap_{i}"])
        G = hq.shape[1]
        gc = np.arange(0, G + 2, dtype=np.float64)
        scen = hip.qm_adjust_linear2d(sim, af, hq, gc, newg, extrap, "+", cubic=method == "cubic")
        exp = sim + d[f"out_{i}"]  # the golden vector is the interpolated factor; kind "+" applies it
        # values to float32 rounding, the NaN pattern (queries outside the hull of the nodes) identical: NO element is let off.
        # (Round 2 allowed 0.2 % of unexplained outliers here; measured on the device path: 0 NaN mismatches, 0 values beyond the
        # tolerance, in every one of the eight cases — scripts/diag/interp2d_mismatch.py.)
        nan_mismatch = np.isnan(scen) != np.isnan(exp)
        both = ~np.isnan(scen) & ~np.isnan(exp)
        tol = 3e-6 if method == "linear" else 2e-5
        bad = ~np.isclose(scen[both], exp[both], rtol=tol, atol=tol * 10)
        assert nan_mismatch.sum() == 0 and bad.sum() == 0, (i, method, extrap, int(nan_mismatch.sum()), int(bad.sum()))
        seen += 1
    assert seen == 4


def test_window_group_bookkeeping_hand_case(hip, golden):
    d = golden("window_groups_handcase")
    x = d["x"][None, :]
    assert d["plain"][0] == 15.0
    np.testing.assert_allclose(hip.group_means(x, d["gidx"], 12, 1)[0], d["plain"], rtol=1e-14)
    np.testing.assert_allclose(hip.group_means(x, d["gidx"], 12, 5)[0], d["win5"], rtol=1e-14)
    np.testing.assert_allclose(hip.group_means(x, d["doy_gidx"], 366, 5)[0], d["doy_win5"], rtol=1e-14)
    x32 = x.astype(np.float32)
    med = hip.quantile_groups(x32, d["doy_gidx"], 366, 5, np.array([0.5], np.float32))[0, :, 0]
    np.testing.assert_array_equal(med[2:-2], x32[0, 2:-2])
    med12 = hip.quantile_groups(x32, d["gidx"], 12, 1, np.array([0.5], np.float32))[0, :, 0]
    np.testing.assert_allclose(med12, d["plain"], rtol=1e-6)  # arange: the monthly median equals the monthly mean
