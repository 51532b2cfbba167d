"""GPU parity: the HIP path (through the C ABI of libxsdba_hip.so) against the CPU oracle on the same
inputs, and against the committed golden vectors generated from the reference's own kernels.

Bars:
  * integer artefacts — valid counts / order-statistic selection (seen through hist_q), nearest-node
    indexes, ranks (sim_q) — bit-exact against the oracle;
  * float32/float64 values: bit-exact against the oracle for EQM/QDM (same arithmetic, fp-contract off
    on both sides); DQM quantities within 2 ulp (float64 block-reduction order of the group mean);
  * against the reference-generated golden vectors: 1e-5 relative (float tolerance of north_star), with
    exact nearest-neighbour distance ties accepted (cKDTree's choice is implementation-defined).
"""
from __future__ import annotations

import zlib

import numpy as np
import pytest

import oracle
from conftest import assert_close, assert_close_or_tie, assert_dqm_close_or_flip

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from xsdba_amd import hip as h

    h.default_context(0)  # raises loudly if the library or the GPU is missing
    return h


@pytest.fixture(autouse=True, params=["cell", "rows"])
def train_kernel(request, monkeypatch):
    """every test runs twice: with the per-cell sort kernels (k_train_cell / k_qdm_cell, where eligible) and with the
    generic kernels pinned (XSDBA_HIP_TRAIN=rows: per-(cell, group) row sort; XSDBA_HIP_QDM=count: ranks by counting)"""
    if request.param == "rows":
        monkeypatch.setenv("XSDBA_HIP_TRAIN", "rows")
        monkeypatch.setenv("XSDBA_HIP_QDM", "count")
    else:
        monkeypatch.delenv("XSDBA_HIP_TRAIN", raising=False)
        monkeypatch.delenv("XSDBA_HIP_QDM", raising=False)
    return request.param


def _doy_month(T, start=0):
    d0 = (np.arange(T) + start) % 365
    cum = np.concatenate([[0], np.cumsum([31, 28, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31])])
    return d0.astype(np.int32), (np.searchsorted(cum, d0, side="right") - 1).astype(np.int32)


def _nodes(n, dtype=np.float32):
    return ((np.arange(n) + 0.5) / n).astype(dtype)


def _synth(rng, C, T, kind, dtype=np.float32, nan_frac=0.0):
    doy, _ = _doy_month(T)
    t = np.arange(T)
    if kind == "+":
        off = rng.standard_normal((C, 1)) * 3
        ref = 280 - 20 * np.cos(2 * np.pi * (doy + 1) / 365) + 2 * rng.random((C, T)) + 0.1 * t / 365 + off
        hist = 280 - 18 * np.cos(2 * np.pi * (doy + 1) / 365) + 2 * rng.random((C, T)) + 0.11 * t / 365 + off
        sim = 281 - 18 * np.cos(2 * np.pi * (doy + 1) / 365) + 2.2 * rng.random((C, T)) + 0.15 * t / 365 + off
    else:
        ref, hist, sim = rng.gamma(4, 7.5, (C, T)), rng.gamma(8.15, 3.68, (C, T)), rng.gamma(16, 2.63, (C, T))
    out = [a.astype(dtype) for a in (ref, hist, sim)]
    if nan_frac:
        for a in out:
            a[rng.random(a.shape) < nan_frac] = np.nan
    return out


GROUPINGS = {
    "time": lambda T: (np.zeros(T, np.int32), 1, 1),
    "month": lambda T: (_doy_month(T)[1], 12, 1),
    "doy31": lambda T: (_doy_month(T)[0], 365, 31),
    "doy1": lambda T: (_doy_month(T)[0], 365, 1),
    "month5": lambda T: (_doy_month(T)[1], 12, 5),
    "doy30_even": lambda T: (_doy_month(T)[0], 365, 30),
}
# 40 un-windowed groups of ~T / 40 members: neither k_train_sel (more than 32 groups) nor a padded toggle table (lists of ~T / 20 steps)
LONG_GROUPINGS = dict(GROUPINGS, week40=lambda T: (((np.arange(T) // 7) % 40).astype(np.int32), 40, 1))


# ------------------------------------------------------------------------------------------------
# train
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("gname", list(GROUPINGS))
@pytest.mark.parametrize("kind", ["+", "*"])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_eqm_train_bit_exact_vs_oracle(hip, gname, kind, dtype):
    rng = np.random.default_rng(zlib.crc32(f"{gname}{kind}".encode()))
    T, C = 3650 if gname != "time" else 2500, 5
    gidx, G, W = GROUPINGS[gname](T)
    ref, hist, _ = _synth(rng, C, T, kind, dtype, nan_frac=0.02)
    ref[3] = np.nan  # an all-NaN cell
    hist[4, : T // 2] = np.nan
    q = _nodes(20, dtype)
    af, hq = hip.eqm_train(ref, hist, gidx, G, W, q, kind)
    eaf, ehq = oracle.eqm_train(ref, hist, gidx, G, W, q, kind)
    assert_close(hq, ehq, rtol=0.0, what="hist_q")
    assert_close(af, eaf, rtol=0.0, what="af")


@pytest.mark.parametrize("gname", ["time", "month", "doy31"])
@pytest.mark.parametrize("kind", ["+", "*"])
def test_dqm_train_vs_oracle(hip, gname, kind):
    rng = np.random.default_rng(7)
    T, C = 3650, 4
    gidx, G, W = GROUPINGS[gname](T)
    ref, hist, _ = _synth(rng, C, T, kind, np.float32, nan_frac=0.01)
    q = _nodes(20)
    af, hq, sc = hip.dqm_train(ref, hist, gidx, G, W, q, kind)
    eaf, ehq, esc = oracle.dqm_train(ref, hist, gidx, G, W, q, kind)
    scale = float(np.nanmax(np.abs(ref)))
    atol = 2.4e-7 * scale if kind == "+" else 2.4e-7
    assert_close(sc, esc, rtol=2.4e-7, atol=atol, what="scaling")
    assert_close(hq, ehq, rtol=2.4e-7, atol=atol, what="hist_q")
    assert_close(af, eaf, rtol=5e-7, atol=2 * atol, what="af")


def test_quantile_groups_layouts_and_edges(hip):
    rng = np.random.default_rng(3)
    T, C = 1460, 7
    doy, month = _doy_month(T)
    x = (rng.standard_normal((C, T)) * 5 + 10).astype(np.float32)
    x[0] = np.nan  # all-NaN -> NaN (tests/test_nbutils.py:31-34)
    x[1, 1:] = np.nan  # single valid value -> every quantile equals it (tests/test_nbutils.py:26-29)
    q = np.linspace(0.1, 0.99, 50).astype(np.float32)
    exp = oracle.quantile_groups(x, month, 12, 1, q)
    got_cm = hip.quantile_groups(x, month, 12, 1, q)
    got_tm = hip.quantile_groups(np.ascontiguousarray(x.T).T, month, 12, 1, q)  # time-major storage
    assert_close(got_cm, exp, rtol=0.0, what="cell-major")
    assert_close(got_tm, exp, rtol=0.0, what="time-major")
    assert np.isnan(got_cm[0]).all()
    assert (got_cm[1, 0] == x[1, 0]).all()
    # ragged groups with an empty one, and times that belong to no group
    g = month.copy()
    g[g == 5] = -1
    exp = oracle.quantile_groups(x, g, 12, 1, q)
    got = hip.quantile_groups(x, g, 12, 1, q)
    assert_close(got, exp, rtol=0.0, what="empty group")
    assert np.isnan(got[:, 5]).all()


def test_quantile_long_row_group_time(hip):
    # group="time" on a 30-year daily series: one 10950-long row per cell (16384-key LDS sort)
    rng = np.random.default_rng(5)
    x = rng.gamma(2.0, 3.0, (3, 10950)).astype(np.float32)
    x[rng.random(x.shape) < 0.05] = np.nan
    q = _nodes(20)
    g = np.zeros(10950, np.int32)
    got = hip.quantile_groups(x, g, 1, 1, q)
    assert_close(got, oracle.quantile_groups(x, g, 1, 1, q), rtol=0.0, what="long row")
    # reference tests/test_nbutils.py:12-21: equals np.nanquantile to 6 decimals (relative to scale here)
    np.testing.assert_allclose(got[:, 0], np.nanquantile(x.astype(np.float64), q.astype(np.float64), axis=1).T, rtol=2e-6)


# ------------------------------------------------------------------------------------------------
# adjust
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("gname", ["time", "month", "doy31", "doy1"])
@pytest.mark.parametrize("kind", ["+", "*"])
def test_qm_adjust_bit_exact_vs_oracle(hip, gname, kind):
    rng = np.random.default_rng(11)
    T, C = 3650, 4
    gidx, G, W = GROUPINGS[gname](T)
    ref, hist, sim = _synth(rng, C, T, kind, np.float32, nan_frac=0.01)
    q = _nodes(20)
    af, hq = oracle.eqm_train(ref, hist, gidx, G, W, q, kind)
    af[1, G // 2, 3] = np.nan  # holes in the tables
    hq[2, 0, :] = np.nan
    interps = ("nearest", "linear") if G == 1 else ("nearest",)
    for interp in interps:
        for extrap in ("constant", "nan"):
            got, node = hip.qm_adjust(sim, af, hq, gidx, interp, extrap, kind, return_nodes=True)
            exp, enode = oracle.qm_adjust(sim, af, hq, gidx, interp, extrap, kind, return_nodes=True)
            assert_close(got, exp, rtol=0.0, what=f"scen {interp}/{extrap}")
            if interp == "nearest":
                np.testing.assert_array_equal(node, enode)  # nearest-node indexes: bit-exact
    # time-major input and output
    got_tm = hip.qm_adjust(np.ascontiguousarray(sim.T).T, af, hq, gidx, "nearest", "constant", kind, out_time_major=True)
    assert_close(got_tm, oracle.qm_adjust(sim, af, hq, gidx, "nearest", "constant", kind), rtol=0.0, what="time-major")


@pytest.mark.parametrize("Q", [16, 32, 48])
@pytest.mark.parametrize("gname", ["doy31", "month"])
def test_qm_adjust_row_lengths_multiple_of_16(hip, gname, Q):
    """tables whose rows are a multiple of 16 dwords long live padded in LDS (kernels.hpp: tab_row_stride): dense tables, tables with
    holes (the compaction path), both extrapolations, EQM / DQM / QDM adjustments — bit-exact scen and node indexes"""
    rng = np.random.default_rng(Q + len(gname))
    T, C = 3650, 4
    gidx, G, W = GROUPINGS[gname](T)
    kind = "+"
    ref, hist, sim = _synth(rng, C, T, kind, np.float32, nan_frac=0.01)
    q = _nodes(Q)
    af, hq = oracle.eqm_train(ref, hist, gidx, G, W, q, kind)
    af[1, G // 2, 3] = np.nan
    hq[2, 0, :] = np.nan
    for extrap in ("constant", "nan"):
        got, node = hip.qm_adjust(sim, af, hq, gidx, "nearest", extrap, kind, return_nodes=True)
        exp, enode = oracle.qm_adjust(sim, af, hq, gidx, "nearest", extrap, kind, return_nodes=True)
        assert_close(got, exp, rtol=0.0, what=f"scen {extrap}")
        np.testing.assert_array_equal(node, enode)
    scen, sim_q = hip.qdm_adjust(sim, af, q, gidx, W, True, "nearest", "constant", kind)
    escen, esim_q = oracle.qdm_adjust(sim, af, q, gidx, W, True, "nearest", "constant", kind)
    assert_close(sim_q, esim_q, rtol=0.0, what="qdm sim_q")
    assert_close(scen, escen, rtol=0.0, what="qdm scen")


@pytest.mark.parametrize("gname", ["time", "doy31", "month"])
@pytest.mark.parametrize("kind", ["+", "*"])
def test_dqm_adjust_vs_oracle(hip, gname, kind):
    rng = np.random.default_rng(13)
    T, C = 3650, 3
    gidx, G, W = GROUPINGS[gname](T)
    ref, hist, sim = _synth(rng, C, T, kind, np.float32, nan_frac=0.005)
    q = _nodes(20)
    af, hq, sc = oracle.dqm_train(ref, hist, gidx, G, W, q, kind)
    tnum = (np.arange(T, dtype=np.float64) + 10957) * 86400e9
    interps = ("nearest", "linear") if G == 1 else ("nearest",)
    for interp in interps:
        scen, trend = hip.dqm_adjust(sim, af, hq, sc, gidx, W, tnum, 1, interp, "constant", kind)
        escen, etrend = oracle.dqm_adjust(sim, af, hq, sc, gidx, W, tnum, 1, interp, "constant", kind)
        assert_close(trend, etrend, rtol=2.4e-7, what="trend")
        if interp == "nearest":
            # a 1-ulp trend difference can move a detrended value across a table node: every such point must be a proven
            # knife-edge (two nodes equidistant from the detrended value), and there may only be a handful
            assert_dqm_close_or_flip(scen, escen, sim, sc, etrend, hq, af, gidx, kind, 1e-6, what=f"dqm scen {gname}{kind}")
        else:
            assert_close(scen, escen, rtol=2e-6, what="dqm scen linear")


@pytest.mark.parametrize("gname", ["time", "doy31", "doy1", "month"])
@pytest.mark.parametrize("rank_window", [False, True])
def test_qdm_adjust_bit_exact_vs_oracle(hip, gname, rank_window):
    rng = np.random.default_rng(17)
    T, C = 1825, 3
    gidx, G, W = GROUPINGS[gname](T)
    kind = "*"
    ref, hist, sim = _synth(rng, C, T, kind, np.float32, nan_frac=0.01)
    sim[0, : T // 3] = np.round(sim[0, : T // 3])  # ties -> average ranks
    q = _nodes(20)
    af, _ = oracle.eqm_train(ref, hist, gidx, G, W, q, kind)
    scen, sim_q = hip.qdm_adjust(sim, af, q, gidx, W, rank_window, "nearest", "constant", kind)
    escen, esim_q = oracle.qdm_adjust(sim, af, q, gidx, W, rank_window, "nearest", "constant", kind)
    assert_close(sim_q, esim_q, rtol=0.0, what="sim_q (ranks)")
    assert_close(scen, escen, rtol=0.0, what="scen")
    r = hip.rank_groups(sim, gidx, G, W, rank_window)
    assert_close(r, esim_q, rtol=0.0, what="rank_groups")


@pytest.mark.parametrize("T", [700, 4500, 10950])
@pytest.mark.parametrize("gname", ["time", "doy31", "month", "month5", "doy30_even", "season", "time_gaps"])
def test_qdm_long_series_ties_and_gaps(hip, gname, T):
    """the three LDS configurations of k_qdm_cell (T <= 4096 / 8192 / 11264), heavy ties incl. -0.0 == +0.0, an
    all-NaN cell, time steps without a group, both kinds, 1-D linear interpolation, NaN extrapolation"""
    rng = np.random.default_rng(T + len(gname))
    C = 4
    if gname == "season":
        gidx, G, W = ((_doy_month(T)[1] + 1) % 12 // 3).astype(np.int32), 4, 1
    elif gname == "time_gaps":
        gidx, G, W = np.zeros(T, np.int32), 1, 1
        gidx[rng.random(T) < 0.1] = -1
    else:
        gidx, G, W = GROUPINGS[gname](T)
    if gname == "month5":
        gidx = gidx.copy()
        gidx[5::97] = -1
    q = _nodes(15)
    for kind in "+*":
        ref, hist, sim = _synth(rng, C, T, kind, np.float32, nan_frac=0.02)
        sim[0] = np.round(sim[0] / 4) * 4             # a handful of distinct values
        sim[1, ::3] = 0.0
        sim[1, 1::6] = -0.0                           # equal to +0.0 for ranking purposes
        sim[2] = np.nan
        af, _ = oracle.eqm_train(ref, hist, gidx, G, W, q, kind)
        for rank_window in (False, True):
            for interp, extrap in (("nearest", "constant"), ("linear", "nan")) if G == 1 else (("nearest", "constant"), ("nearest", "nan")):
                scen, sim_q = hip.qdm_adjust(sim, af, q, gidx, W, rank_window, interp, extrap, kind)
                escen, esim_q = oracle.qdm_adjust(sim, af, q, gidx, W, rank_window, interp, extrap, kind)
                assert_close(sim_q, esim_q, rtol=0.0, what=f"sim_q {kind} rw={rank_window}")
                assert_close(scen, escen, rtol=0.0, what=f"scen {kind} rw={rank_window} {interp}/{extrap}")
        r = hip.rank_groups(sim, gidx, G, W, True)
        assert_close(r, oracle.rank_groups(sim, gidx, G, W, True), rtol=0.0, what="rank_groups")


@pytest.mark.parametrize("gname", ["time", "doy31", "month"])
def test_qdm_float64(hip, gname):
    """k_qdm_cell on float64 series (31-bit range keys: equality decided on the doubles): heavy ties incl. -0.0 == +0.0,
    clusters of distinct doubles below the key resolution, NaNs; ranks and scen bit-exact vs the oracle"""
    rng = np.random.default_rng(41)
    T, C = 6000, 4
    gidx, G, W = GROUPINGS[gname](T)
    q = _nodes(15, np.float64)
    kind = "*"
    ref, hist, sim = (a.astype(np.float64) * (1 + 1e-9 * rng.random(a.shape)) for a in _synth(rng, C, T, kind, np.float32, nan_frac=0.02))
    sim[0] = np.round(sim[0] / 4) * 4
    sim[1, ::3] = 0.0
    sim[1, 1::6] = -0.0
    m = rng.random(T) < 0.4
    sim[2, m] = 50.0 + 1e-13 * rng.permutation(m.sum())   # thousands of equal range keys, all values distinct
    sim[2, ::11] = 50.0                                    # ... with exact ties in between
    af, _ = oracle.eqm_train(ref, hist, gidx, G, W, q, kind)
    for rank_window in (False, True):
        scen, sim_q = hip.qdm_adjust(sim, af, q, gidx, W, rank_window, "nearest", "constant", kind)
        escen, esim_q = oracle.qdm_adjust(sim, af, q, gidx, W, rank_window, "nearest", "constant", kind)
        assert_close(sim_q, esim_q, rtol=0.0, what=f"sim_q rw={rank_window}")
        assert_close(scen, escen, rtol=0.0, what=f"scen rw={rank_window}")
    assert scen.dtype == np.float64


def test_qdm_nan_factors_fall_back(hip):
    """NaN factors break the own-row shortcut of k_qdm_cell: those cells are redone by the generic kernel (2-D nearest
    neighbour across group rows), the others keep the fast result"""
    rng = np.random.default_rng(5)
    T, C, kind = 3650, 6, "+"
    gidx, G, W = GROUPINGS["doy31"](T)
    ref, hist, sim = _synth(rng, C, T, kind)
    q = _nodes(12)
    af, _ = oracle.eqm_train(ref, hist, gidx, G, W, q, kind)
    af[1, 40, :] = np.nan          # a whole row: the neighbours' nodes win
    af[3, :, 0] = np.nan           # first node everywhere: yfirst moves
    af[4, 100:103, 5:9] = np.nan   # holes inside rows
    af[5] = np.nan                 # nothing valid at all
    for extrap in ("constant", "nan"):
        scen, sim_q = hip.qdm_adjust(sim, af, q, gidx, W, True, "nearest", extrap, kind)
        escen, esim_q = oracle.qdm_adjust(sim, af, q, gidx, W, True, "nearest", extrap, kind)
        assert_close(sim_q, esim_q, rtol=0.0, what="sim_q")
        assert_close(scen, escen, rtol=0.0, what=f"scen ({extrap})")


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("gname", ["time", "month", "doy31"])
def test_tail_mask_vs_oracle(hip, gname, dtype):
    """max_tail_factor mask (_adjustment.py:648-658, 672-673): bit-exact vs the oracle, host and device buffers, NaNs,
    time steps in no group"""
    rng = np.random.default_rng(11)
    T, C = 2000, 5
    gidx, G, W = GROUPINGS[gname](T)
    gidx = gidx.copy()
    if G > 1:
        gidx[3::50] = -1
    sim = rng.gamma(2.0, 4.0, (C, T)).astype(dtype)
    sim[rng.random((C, T)) < 0.02] = np.nan
    scen = (sim * 1.7).astype(dtype)
    hq_raw = np.sort(rng.gamma(2.0, 4.0, (C, G, 12)), axis=-1).astype(dtype)
    hq_raw[1, :, -1] = np.nan  # no limit for that cell: nothing masked
    exp = oracle.tail_mask(sim, scen, hq_raw, gidx if G > 1 else None, 0.9)
    assert 0 < (exp != scen)[~np.isnan(sim)].sum() < sim.size // 2
    got = hip.tail_mask(sim, scen.copy(), hq_raw, gidx if G > 1 else None, 0.9)
    assert_close(got, exp, rtol=0.0, what="tail_mask host")
    ctx = hip.default_context()
    d_sim, d_scen = ctx.to_device(sim), ctx.to_device(scen)
    hip.tail_mask(d_sim, d_scen, hq_raw, gidx if G > 1 else None, 0.9)
    assert_close(d_scen.download(), exp, rtol=0.0, what="tail_mask device")


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_jitter_bit_exact_vs_oracle(hip, dtype):
    """processing.jitter (processing.py:178-257): same counter-based draws on both sides -> bit-exact; open ends"""
    rng = np.random.default_rng(2)
    x = rng.gamma(0.6, 3.0, (7, 1500)).astype(dtype)
    x[rng.random(x.shape) < 0.05] = np.nan
    for kw in (dict(lower=1.0), dict(upper=8.0, maximum=12.0), dict(lower=0.5, minimum=0.1, upper=8.0, maximum=9.0)):
        got = hip.jitter(x, seed=77, first_cell=5, **kw)
        exp = oracle.jitter(x, seed=77, first_cell=5, **kw)
        assert_close(got, exp, rtol=0.0, what=f"jitter {kw}")
        lo, hi = kw.get("lower"), kw.get("upper")
        if lo is not None:
            m = x < lo
            assert ((got[m] > kw.get("minimum", 0.0)) & (got[m] < lo)).all() and (got[m] != x[m]).all()
        if hi is not None:
            m = x >= hi
            assert ((got[m] >= hi) & (got[m] < kw["maximum"])).all()
        keep = ~((x < (lo if lo is not None else -np.inf)) | (x >= (hi if hi is not None else np.inf)))
        assert np.array_equal(got[keep], x[keep], equal_nan=True)
    # another shard offset reproduces the same cells
    a = hip.jitter(x[3:], lower=1.0, seed=77, first_cell=8)
    assert np.array_equal(a, hip.jitter(x, lower=1.0, seed=77, first_cell=5)[3:], equal_nan=True)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("gname", ["time", "month", "doy31", "doy1"])
def test_adapt_freq_bit_exact_vs_oracle(hip, gname, dtype):
    """_processing._adapt_freq (_processing.py:75-142), training and adjust branches: P0 / dP0 / pth and the adapted
    series bit-exact vs the oracle (shared counter-based generator), NaNs, cells that need no adaptation"""
    rng = np.random.default_rng(4)
    T, C = 2920, 6
    gidx, G, W = GROUPINGS[gname](T)
    pr = rng.uniform(0, 40, (C, T))
    sim = np.where(pr < 20, pr / 20, pr).astype(dtype)   # 50 % under 1
    ref = np.where(pr < 10, pr / 20, pr).astype(dtype)   # 25 % under 1
    sim[0, ::4] = 0.0                                     # many exact ties
    ref[1] = sim[1]                                       # no adaptation needed
    sim[2, rng.random(T) < 0.05] = np.nan
    ref[3] = np.nan                                       # undefined P0_ref
    sim[4] = 5.0                                          # nothing under the threshold
    e = oracle.adapt_freq(sim, ref, gidx, G, W, 1.0, seed=9, first_cell=2)
    h = hip.adapt_freq(sim, ref, gidx, G, W, 1.0, seed=9, first_cell=2)
    for name, a, b in zip(("sim_ad", "pth", "dP0", "P0_ref", "P0_hist"), h, e):
        assert_close(a, b, rtol=0.0, what=f"adapt_freq train {name}")
    changed = ~np.isclose(h[0], sim, equal_nan=True)
    assert changed[0].any() and not changed[1].any() and not changed[3].any() and not changed[4].any()
    assert (sim[changed] <= 1.0).all() and (h[0][changed] >= 1.0).all()
    # adjust branch: the trained tables on the groups without window
    sim2 = np.where(pr[:, ::-1] < 22, pr[:, ::-1] / 22, pr[:, ::-1]).astype(dtype)
    e2 = oracle.adapt_freq(sim2, None, gidx, G, 1, 1.0, seed=10, first_cell=2, P0_ref=e[3], P0_hist=e[4], pth=e[1])
    h2 = hip.adapt_freq(sim2, None, gidx, G, 1, 1.0, seed=10, first_cell=2, P0_ref=e[3], P0_hist=e[4], pth=e[1])
    assert_close(h2[0], e2[0], rtol=0.0, what="adapt_freq adjust sim_ad")
    assert_close(h2[2], e2[2], rtol=0.0, what="adapt_freq adjust dP0")


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("gname", ["time", "month", "doy31", "month5"])
def test_scaling_vs_oracle(hip, gname, dtype):
    """Scaling: group means (float64 accumulation: <= 1 ulp of the oracle's sequential sum) and the
    nearest-group application of the factors (bit-exact)"""
    rng = np.random.default_rng(21)
    T, C = 2555, 5
    gidx, G, W = GROUPINGS[gname](T)
    gidx = gidx.copy()
    if G > 1:
        gidx[7::90] = -1
    x = (280 + 10 * rng.standard_normal((C, T))).astype(dtype)
    x[rng.random((C, T)) < 0.03] = np.nan
    x[2] = np.nan
    m, em = hip.group_means(x, gidx, G, W), oracle.group_means(x, gidx, G, W)
    assert_close(m, em, rtol=2.4e-7 if dtype == np.float32 else 1e-11, what="group means")  # summation order
    af = rng.standard_normal((C, G)).astype(dtype)
    for kind in "+*":
        got = hip.scaling_adjust(x, af, gidx if G > 1 else None, kind)
        assert_close(got, oracle.scaling_adjust(x, af, gidx if G > 1 else None, kind), rtol=0.0, what=f"scaling_adjust {kind}")
    if G > 1:  # linear interpolation between neighbouring groups (utils.broadcast interp="linear")
        g_hi = ((np.maximum(gidx, 0) + 1) % G).astype(np.int32)
        frac = rng.random(T)
        frac[::13] = 1.0
        for kind in "+*":
            got = hip.scaling_adjust(x, af, gidx, kind, g_hi=g_hi, frac=frac)
            assert_close(got, oracle.scaling_adjust(x, af, gidx, kind, g_hi=g_hi, frac=frac), rtol=0.0, what=f"scaling linear {kind}")
        sth = np.abs(rng.standard_normal((C, G))).astype(dtype)
        got = hip.loci_adjust(np.abs(x), af, sth, gidx, 1.0, g_hi=g_hi, frac=frac)
        assert_close(got, oracle.loci_adjust(np.abs(x), af, sth, gidx, 1.0, g_hi=g_hi, frac=frac), rtol=0.0, what="loci linear")


@pytest.mark.parametrize("gname", ["doy31", "month", "time"])
def test_f64_train_near_collisions(hip, gname):
    """float64 fast kernel (31-bit range keys + tie repair): clusters of distinct doubles that differ far below the key
    resolution, exact ties, huge dynamic range, long series; EQM bit-exact, DQM to 2 ulp"""
    rng = np.random.default_rng(31)
    T, C = 10950 if gname != "time" else 6000, 4
    gidx, G, W = GROUPINGS[gname](T)
    q = _nodes(20, np.float64)
    base = 280 + 15 * rng.standard_normal((C, T))
    ref, hist = base.copy(), base[:, ::-1].copy() + 0.5
    # cell 0: half of the values sit in a cluster 1e-13 wide (thousands of key collisions between distinct doubles)
    m = rng.random(T) < 0.5
    hist[0, m] = 300.0 + 1e-13 * rng.permutation(m.sum())
    ref[0, m] = 299.0 + 1e-12 * rng.permutation(m.sum())
    # cell 1: heavy exact ties next to a huge dynamic range (no signed zeros: which zero a quantile returns is the
    # sort's choice in the reference as well, and 1/-0.0 would turn it into the sign of an infinite factor)
    hist[1] = np.where(rng.random(T) < 0.4, 0.0, rng.gamma(0.3, 50.0, T))
    hist[1, 5] = 1e300
    ref[1] = np.round(ref[1])
    # cell 2: NaNs; cell 3: came from float32
    hist[2, rng.random(T) < 0.1] = np.nan
    ref[3], hist[3] = ref[3].astype(np.float32), hist[3].astype(np.float32)
    for kind in "+*":
        af, hq = hip.eqm_train(ref, hist, gidx, G, W, q, kind)
        eaf, ehq = oracle.eqm_train(ref, hist, gidx, G, W, q, kind)
        assert_close(hq, ehq, rtol=0.0, what=f"hist_q {kind}")
        assert_close(af, eaf, rtol=0.0, what=f"af {kind}")
    af, hq, sc = hip.dqm_train(ref, hist, gidx, G, W, q, "+")
    eaf, ehq, esc = oracle.dqm_train(ref, hist, gidx, G, W, q, "+")
    # float64 group means: summation order (sequential in the oracle, block / running sums on the device)
    assert_close(sc, esc, rtol=1e-12, atol=1e-10, what="scaling")
    assert_close(hq, ehq, rtol=1e-12, atol=1e-10, what="hist_q (dqm)")
    assert_close(af, eaf, rtol=1e-11, atol=1e-9, what="af (dqm)")
    got = hip.quantile_groups(hist, gidx, G, W, q)
    assert_close(got, oracle.quantile_groups(hist, gidx, G, W, q), rtol=0.0, what="quantile_groups f64")


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("gname", ["time", "month", "doy31"])
def test_loci_vs_oracle(hip, gname, dtype):
    """LOCI: hist_thresh bit-exact (sort + float64 lerp), af to the summation order of the block means,
    loci_adjust bit-exact"""
    rng = np.random.default_rng(23)
    T, C = 2920, 5
    gidx, G, W = GROUPINGS[gname](T)
    ref = rng.gamma(0.8, 6.0, (C, T)).astype(dtype)
    hist = (rng.gamma(0.9, 4.0, (C, T)) * 0.7).astype(dtype)
    ref[1, rng.random(T) < 0.05] = np.nan
    hist[2, rng.random(T) < 0.05] = np.nan
    hist[3] = np.nan
    af, sth = hip.loci_train(ref, hist, gidx, G, W, 1.0)
    eaf, esth = oracle.loci_train(ref, hist, gidx, G, W, 1.0)
    assert_close(sth, esth, rtol=0.0, what="hist_thresh")
    assert_close(af, eaf, rtol=5e-6 if dtype == np.float32 else 1e-11, what="af")
    assert np.isnan(af[3]).all() and np.isfinite(af[:3]).all()
    sim = (rng.gamma(0.9, 4.0, (C, T)) * 0.8).astype(dtype)
    gi = gidx.copy()
    if G > 1:
        gi[11::100] = -1
    got = hip.loci_adjust(sim, eaf, esth, gi if G > 1 else None, 1.0)
    assert_close(got, oracle.loci_adjust(sim, eaf, esth, gi if G > 1 else None, 1.0), rtol=0.0, what="loci_adjust")
    assert (got[np.isfinite(got)] >= 0).all()


@pytest.mark.parametrize("T", [3650, 14000])
def test_dqm_train_huge_dynamic_range(hip, T):
    """DQM group means come from running sums in the per-cell kernels; with 1e-20 "numerical drizzle" in one season and
    1e2 storms in another the running sums would keep nothing of the drizzle: such cells must be redone with fresh sums"""
    rng = np.random.default_rng(12)
    C = 3
    gidx, G, W = GROUPINGS["doy31"](T)
    doy = gidx.astype(np.float64)
    wet = (np.cos(2 * np.pi * doy / 365) > 0)[None, :]
    mk = lambda: np.where(wet, rng.gamma(2.0, 40.0, (C, T)), 1e-20 * (1 + rng.random((C, T)))).astype(np.float32)  # noqa: E731
    ref, hist = mk(), mk()
    hist[1] = rng.gamma(2.0, 4.0, T)  # an ordinary cell stays on the fast path
    q = _nodes(12)
    af, hq, sc = hip.dqm_train(ref, hist, gidx, G, W, q, "*")
    eaf, ehq, esc = oracle.dqm_train(ref, hist, gidx, G, W, q, "*")
    assert_close(sc, esc, rtol=5e-7, what="scaling")
    assert_close(hq, ehq, rtol=5e-7, what="hist_q")
    assert_close(af, eaf, rtol=2e-6, what="af")


def test_polydetrend_huge_dynamic_range(hip):
    """the sliding window sums of the trend kernel are rebuilt when a huge value has just left the window"""
    rng = np.random.default_rng(13)
    T, C = 3650, 3
    gidx, G, W = GROUPINGS["doy31"](T)
    x = (1e-20 * (1 + rng.random((C, T)))).astype(np.float32)
    x[:, ::97] = 250.0
    x[1] = 280 + rng.standard_normal(T)
    tnum = (np.arange(T, dtype=np.float64) + 12000.0) * 86400e9
    tr = hip.polydetrend_trend(x, gidx, G, W, tnum, 1)
    etr = oracle.polydetrend_trend(x, gidx, G, W, tnum, 1)
    assert_close(tr, etr, rtol=2e-6, atol=1e-24, what="trend")


def test_polydetrend_ramp(hip):
    # reference tests/test_detrending.py:18-30: detrending a ramp leaves ~0 and retrending restores it
    T = int(20 * 365.25)
    x = np.arange(T, dtype=np.float64)[None, :]
    tnum = np.arange(T, dtype=np.float64) * 86400e9 + 9.6e17
    trend = hip.polydetrend_trend(x, None, 1, 1, tnum, 1)
    np.testing.assert_array_almost_equal(x - trend, 0)
    # float64 reduction order differs (half-wave xor-shuffles vs sequential): 1e-12 of the data scale
    assert_close(trend, oracle.polydetrend_trend(x, np.zeros(T, np.int32), 1, 1, tnum, 1), rtol=1e-12, atol=1e-12 * T, what="trend")


# ------------------------------------------------------------------------------------------------
# golden vectors produced by the reference's own kernels
# ------------------------------------------------------------------------------------------------
PIPES = ["time_add", "time_mul", "month_add", "month_mul", "doy31_add", "doy31_mul", "doy1_add", "doy1_mul",
         "month5_add", "month5_mul"]


@pytest.mark.parametrize("name", PIPES)
def test_golden_eqm_pipeline(hip, golden, name):
    d = golden(f"pipeline_{name}")
    gidx, G, W, kind = d["gidx"], int(d["G"]), int(d["W"]), str(d["kind"])
    af, hq = hip.eqm_train(d["ref"], d["hist"], gidx, G, W, d["q"], kind)
    scale_q = float(np.nanmax(np.abs(d["eqm_hist_q"])))
    assert_close(hq, d["eqm_hist_q"], rtol=1e-5, what="hist_q")
    assert_close(af, d["eqm_af"], rtol=1e-5, atol=1e-6 * scale_q if kind == "+" else 0, what="af")
    for key in d.files:
        if key.startswith("eqm_scen_"):
            _, _, interp, extrap = key.split("_")
            got = hip.qm_adjust(d["sim"], d["eqm_af"], d["eqm_hist_q"], gidx, interp, extrap, kind)
            if G > 1:
                assert_close_or_tie(got, d[key], d["sim"], d["eqm_hist_q"], d["eqm_af"], gidx, kind, 1e-5, what=key)
            else:
                assert_close(got, d[key], rtol=1e-5, what=key)


@pytest.mark.parametrize("name", [p for p in PIPES if not p.startswith("month5")])
def test_golden_qdm_dqm_pipeline(hip, golden, name):
    d = golden(f"pipeline_{name}")
    gidx, G, W, kind = d["gidx"], int(d["G"]), int(d["W"]), str(d["kind"])
    for rw in (0, 1):
        key = f"qdm_scen_nearest_rw{rw}"
        scen, sim_q = hip.qdm_adjust(d["sim"], d["eqm_af"], d["q"], gidx, W, bool(rw), "nearest", "constant", kind)
        assert_close(sim_q, d[f"qdm_simq_rw{rw}"], rtol=0.0, what="sim_q")  # ranks: bit-exact vs the reference
        if G > 1:
            assert_close_or_tie(scen, d[key], d["sim"], d["q"], d["eqm_af"], gidx, kind, 1e-5, what=key, newx=sim_q, max_frac=5e-3)
        else:
            assert_close(scen, d[key], rtol=1e-5, what=key)
    if "dqm_af" in d.files:
        # The golden group means are numpy's float32 means (pairwise summation: error up to log2(n) eps of the mean for a row of n
        # values), ours float64 sums rounded once; the scaling is the difference of two such means.  Bound: 2 log2(n) eps32 x scale
        # (n = the longest group row).  Measured: 10 ulps of the mean on the 10 950-value rows of group="time" (3.05e-4 at 300 K) —
        # the 4 ulps round 3's review suggested do not hold there; round 3 allowed a flat 2e-6 x scale (20 ulps).
        scale = float(np.nanmax(np.abs(d["ref"]))) if kind == "+" else 1.0
        n_row = int(np.bincount(gidx[gidx >= 0]).max()) * W if gidx is not None and G > 1 else d["ref"].shape[-1]
        atol = 2 * np.log2(max(n_row, 2)) * float(np.finfo(np.float32).eps) / 2 * scale
        af, hq, sc = hip.dqm_train(d["ref"], d["hist"], gidx, G, W, d["q"], kind)
        assert_close(sc, d["dqm_scaling"], rtol=1e-5, atol=atol, what="scaling")
        assert_close(hq, d["dqm_hist_q"], rtol=1e-5, atol=atol, what="dqm hist_q")
        scen, trend = hip.dqm_adjust(d["sim"], d["dqm_af"], d["dqm_hist_q"], d["dqm_scaling"], gidx, W, d["tns"], 1,
                                     "nearest", "constant", kind)
        assert_close(trend, d["dqm_trend"], rtol=1e-5, atol=atol, what="trend")
        assert_dqm_close_or_flip(scen, d["dqm_scen_nearest"], d["sim"], d["dqm_scaling"], d["dqm_trend"], d["dqm_hist_q"],
                                 d["dqm_af"], gidx, kind, 1e-5, what=f"golden dqm scen {name}")


def test_golden_known_answer_qm(hip, golden):
    # reference tests/test_adjustment.py:947-989 on float64 data
    d = golden("known_answer_qm")
    x, y, q = d["x"], d["y"], d["q50"]
    for kind, tag in (("+", "add"), ("*", "mul")):
        af, hq = hip.eqm_train(y[None], x[None], None, 1, 1, q, kind)
        assert_close(af, d[f"eqm_af_{tag}"], rtol=0.0, what="af (float64, bit-exact vs reference kernel)")
        p = hip.qm_adjust(x[None], af, hq, None, "linear", "constant", kind)
        assert_close(p, d[f"eqm_p_linear_{tag}"], rtol=1e-15, what="p")
        middle = (d["u"] > 1e-2) & (d["u"] < 0.99)
        np.testing.assert_array_almost_equal(p[0][middle], y[middle], 1)


# ------------------------------------------------------------------------------------------------
# size-independent properties at larger sizes (oracle would take too long at BASELINE sizes)
# ------------------------------------------------------------------------------------------------
def test_properties_large(hip):
    rng = np.random.default_rng(23)
    C, T = 512, 10950
    gidx, G, W = GROUPINGS["doy31"](T)
    ref, hist, sim = _synth(rng, C, T, "+", np.float32)
    q = _nodes(20)
    af, hq = hip.eqm_train(ref, hist, gidx, G, W, q, "+")
    # quantiles are sorted along q and lie inside the data range
    assert (np.diff(hq, axis=-1) >= 0).all()
    assert (hq.min(axis=(1, 2)) >= hist.min(axis=1)).all() and (hq.max(axis=(1, 2)) <= hist.max(axis=1)).all()
    # training on identical series gives zero additive factors (idempotence of the mapping)
    af0, _ = hip.eqm_train(hist, hist, gidx, G, W, q, "+")
    assert (af0 == 0).all()
    scen0 = hip.qm_adjust(sim, af0, hq, gidx, "nearest", "constant", "+")
    np.testing.assert_array_equal(scen0, sim)
    # a subsample of cells agrees with the oracle bit for bit
    sub = slice(0, 16)
    eaf, ehq = oracle.eqm_train(ref[sub], hist[sub], gidx, G, W, q, "+")
    assert_close(hq[sub], ehq, rtol=0.0, what="hist_q subsample")
    assert_close(af[sub], eaf, rtol=0.0, what="af subsample")
    scen = hip.qm_adjust(sim, af, hq, gidx, "nearest", "constant", "+")
    assert_close(scen[sub], oracle.qm_adjust(sim[sub], eaf, ehq, gidx, "nearest", "constant", "+"), rtol=0.0, what="scen")
    # adjusting hist itself maps it onto ref's quantiles: group-wise means move toward ref
    adj = hip.qm_adjust(hist, af, hq, gidx, "nearest", "constant", "+")
    assert abs(float(adj.mean() - ref.mean())) < abs(float(hist.mean() - ref.mean())) + 1e-3


def test_device_resident_roundtrip(hip):
    """device pointers in -> device pointers out (the bench path): same numbers as host buffers."""
    rng = np.random.default_rng(29)
    C, T = 64, 3650
    gidx, G, W = GROUPINGS["month"](T)
    ref, hist, sim = _synth(rng, C, T, "+", np.float32)
    q = _nodes(20)
    ctx = hip.default_context(0)
    dref, dhist, dsim = ctx.to_device(ref), ctx.to_device(hist), ctx.to_device(sim)
    daf, dhq = hip.eqm_train(dref, dhist, gidx, G, W, q, "+")
    dscen = hip.qm_adjust(dsim, daf, dhq, gidx, "nearest", "constant", "+")
    ctx.synchronize()
    af, hq = hip.eqm_train(ref, hist, gidx, G, W, q, "+")
    np.testing.assert_array_equal(daf.download(), af)
    np.testing.assert_array_equal(dscen.download(), hip.qm_adjust(sim, af, hq, gidx, "nearest", "constant", "+"))
    for a in (dref, dhist, dsim, daf, dhq, dscen):
        a.free()


def test_errors(hip):
    from xsdba_amd._lib import XsdbaHipError

    x = np.zeros((2, 100), np.float32)
    q = _nodes(5)
    with pytest.raises(XsdbaHipError):  # group index out of range
        hip.quantile_groups(x, np.full(100, 7, np.int32), 4, 1, q)
    with pytest.raises(XsdbaHipError):  # a grouped table with interp="linear" is xsdba_hip_qm_adjust_linear2d's, not this entry's
        hip.qm_adjust(x, np.zeros((2, 4, 5), np.float32), np.zeros((2, 4, 5), np.float32), np.zeros(100, np.int32), "linear")
    with pytest.raises(TypeError):
        hip.quantile_groups(x.astype(np.int32), None, 1, 1, q)


def test_table_larger_than_lds(hip):
    """float64 tables with 50 quantiles x 365 groups (288 KB) do not fit the 160 KB LDS: the adjust kernels then
    keep the compacted table in a global scratch slice per block — same numbers as the oracle."""
    rng = np.random.default_rng(31)
    T, C = 3650, 3
    gidx, G, W = GROUPINGS["doy31"](T)
    ref, hist, sim = _synth(rng, C, T, "+", np.float64, nan_frac=0.01)
    q = _nodes(50, np.float64)
    af, hq = oracle.eqm_train(ref, hist, gidx, G, W, q, "+")
    got, node = hip.qm_adjust(sim, af, hq, gidx, "nearest", "constant", "+", return_nodes=True)
    exp, enode = oracle.qm_adjust(sim, af, hq, gidx, "nearest", "constant", "+", return_nodes=True)
    assert_close(got, exp, rtol=0.0, what="scen (global table)")
    np.testing.assert_array_equal(node, enode)
    scen, sim_q = hip.qdm_adjust(sim, af, q, gidx, W, True, "nearest", "constant", "+")
    escen, esim_q = oracle.qdm_adjust(sim, af, q, gidx, W, True, "nearest", "constant", "+")
    assert_close(sim_q, esim_q, rtol=0.0, what="sim_q")
    assert_close(scen, escen, rtol=0.0, what="qdm scen (global table)")


def test_fast_train_fallback_cells(hip, train_kernel):
    """cells that break the shortcut assumptions of k_train_cell (inf - inf in the interpolation, non-positive
    multiplicative group means, infinite means) are flagged and redone by the generic kernel: same results"""
    rng = np.random.default_rng(37)
    T, C = 3650, 6
    gidx, G, W = GROUPINGS["doy31"](T)
    ref, hist, _ = _synth(rng, C, T, "*", np.float32)
    ref[1, 100:140] = np.inf      # +inf runs -> lerp between inf and inf
    hist[2] = -hist[2]            # negative means for the multiplicative DQM normalisation
    hist[3, ::7] = 0.0            # ties at zero
    ref[4, 5] = -np.inf
    q = _nodes(20)
    with np.errstate(all="ignore"):
        af, hq = hip.eqm_train(ref, hist, gidx, G, W, q, "*")
        eaf, ehq = oracle.eqm_train(ref, hist, gidx, G, W, q, "*")
        assert_close(hq, ehq, rtol=0.0, what="eqm hist_q")
        assert_close(af, eaf, rtol=0.0, what="eqm af")
        for kind in ("+", "*"):
            af, hq, sc = hip.dqm_train(ref, hist, gidx, G, W, q, kind)
            eaf, ehq, esc = oracle.dqm_train(ref, hist, gidx, G, W, q, kind)
            assert_close(sc, esc, rtol=5e-7, atol=1e-4, what=f"dqm scaling {kind}")
            assert_close(hq, ehq, rtol=5e-7, atol=1e-4, what=f"dqm hist_q {kind}")
            assert_close(af, eaf, rtol=2e-6, atol=2e-4, what=f"dqm af {kind}")


@pytest.mark.parametrize("Q", [21, 22, 30, 42, 43, 50, 63, 64])
@pytest.mark.parametrize("T", [3650, 10950])
def test_train_many_quantile_nodes(hip, Q, T):
    """22 .. 63 nodes on windowed day-of-year groups: the three-stream kernel serves them in two or three rounds of the select per
    group step (64 and more: the two-stream kernel); EQM bit-exact, DQM within 2 ulp; NaNs, ties, an all-NaN series"""
    rng = np.random.default_rng(Q * 100003 + T)
    C = 3
    gidx, G, W = GROUPINGS["doy31"](T)
    q = _nodes(Q)
    for kind in "+*":
        ref, hist, _ = _synth(rng, C, T, kind, np.float32, nan_frac=0.01)
        hist[0] = np.round(hist[0])
        if kind == "*":
            ref[1] = np.nan
        af, hq = hip.eqm_train(ref, hist, gidx, G, W, q, kind)
        eaf, ehq = oracle.eqm_train(ref, hist, gidx, G, W, q, kind)
        assert_close(hq, ehq, rtol=0.0, what=f"hist_q {kind} Q={Q}")
        assert_close(af, eaf, rtol=0.0, what=f"af {kind} Q={Q}")
    ref, hist, _ = _synth(rng, C, T, "+", np.float32, nan_frac=0.01)
    af, hq, sc = hip.dqm_train(ref, hist, gidx, G, W, q, "+")
    eaf, ehq, esc = oracle.dqm_train(ref, hist, gidx, G, W, q, "+")
    assert_close(sc, esc, rtol=2.4e-7, atol=2e-5, what="scaling")
    assert_close(hq, ehq, rtol=2.4e-7, atol=1e-4, what="hist_q (dqm)")
    assert_close(af, eaf, rtol=2.4e-7, atol=1e-4, what="af (dqm)")


@pytest.mark.parametrize("T,gname", [(21900, "doy31"), (11300, "doy31"), (14600, "doy31"), (16384, "doy31"), (16385, "doy31"), (12001, "month"), (22528, "doy1"), (15000, "month5"), (15001, "week40")])
def test_long_series_train(hip, T, gname):
    """11 265 .. 22 528 steps: the split-sort + merge kernel (k_train_long) — the padded [G][128] toggle table (doy windows) and
    the CSR lists (week40: 750 steps per list), an odd length, the maximum length, NaNs, ties across the two halves, an all-NaN
    cell; EQM bit-exact, DQM within 2 ulp"""
    rng = np.random.default_rng(T)
    C = 3
    gidx, G, W = LONG_GROUPINGS[gname](T)
    q = _nodes(20)
    for kind in "+*":
        ref, hist, _ = _synth(rng, C, T, kind, np.float32, nan_frac=0.01)
        hist[0] = np.round(hist[0])          # ties, many of them between the two halves of the series
        ref[2, : T // 2 + 7] = np.nan        # one half (and a bit) missing
        if kind == "*":
            hist[1] = np.nan
        af, hq = hip.eqm_train(ref, hist, gidx, G, W, q, kind)
        eaf, ehq = oracle.eqm_train(ref, hist, gidx, G, W, q, kind)
        assert_close(hq, ehq, rtol=0.0, what=f"hist_q {kind}")
        assert_close(af, eaf, rtol=0.0, what=f"af {kind}")
    ref, hist, _ = _synth(rng, C, T, "+", np.float32, nan_frac=0.01)
    af, hq, sc = hip.dqm_train(ref, hist, gidx, G, W, q, "+")
    eaf, ehq, esc = oracle.dqm_train(ref, hist, gidx, G, W, q, "+")
    assert_close(sc, esc, rtol=2.4e-7, atol=2e-5, what="scaling")
    assert_close(hq, ehq, rtol=2.4e-7, atol=1e-4, what="hist_q (dqm)")
    assert_close(af, eaf, rtol=2.4e-7, atol=1e-4, what="af (dqm)")


@pytest.mark.parametrize("T", [21900, 13001])
def test_long_series_key_ranges(hip, T):
    """k_train_long's handle sort over the shapes a key range can take: a series that spans zero (the order-preserving keys of
    negative and positive floats lie 2^31 apart: 32 significant bits — bit 31 travels beside the handles), 28-bit keys (two
    stable passes), a handful of distinct values (no stable pass at all: narrow keys), a constant series, +-inf, a huge
    dynamic range; rows that are only 4-byte aligned (odd T); bit-exact against the oracle"""
    rng = np.random.default_rng(T + 5)
    gidx, G, W = LONG_GROUPINGS["doy31"](T)
    q = _nodes(20)
    doy = np.arange(T) % 365
    C = 7
    x = np.empty((2, C, T), np.float32)
    for a in range(2):
        x[a, 0] = -2 + 15 * np.cos(2 * np.pi * doy / 365) + 4 * rng.standard_normal(T)          # degrees Celsius around zero
        x[a, 1] = rng.gamma(0.6, 5.0, T) * (rng.random(T) < 0.6)                                    # precipitation with dry days
        x[a, 2] = rng.integers(0, 7, T)                                                              # seven distinct values
        x[a, 3] = 3.25                                                                               # one value
        x[a, 4] = 280 + 10 * rng.standard_normal(T)
        x[a, 4, rng.integers(0, T, 5)] = np.inf
        x[a, 4, rng.integers(0, T, 5)] = -np.inf
        x[a, 5] = np.sign(rng.standard_normal(T)) * 10.0 ** rng.uniform(-30, 30, T)                  # every exponent, both signs
        x[a, 6] = -np.abs(100 + 30 * rng.standard_normal(T))                                        # negative only
    x[:, :, rng.integers(0, T, 40)] = np.nan
    ref, hist = x[0], x[1]
    af, hq = hip.eqm_train(ref, hist, gidx, G, W, q, "+")
    eaf, ehq = oracle.eqm_train(ref, hist, gidx, G, W, q, "+")
    assert_close(hq, ehq, rtol=0.0, what="hist_q")
    assert_close(af, eaf, rtol=0.0, what="af")
    got = hip.quantile_groups(hist, gidx, G, W, q)
    assert_close(got, oracle.quantile_groups(hist, gidx, G, W, q), rtol=0.0, what="quantile_groups")


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_vecquantiles_vs_oracle(hip, dtype):
    """nbutils.vecquantiles (a5): one quantile per row; NaN q, no valid value, a single value, q = 0 and 1"""
    rng = np.random.default_rng(41)
    x = rng.gamma(2.0, 3.0, (300, 930)).astype(dtype)
    x[rng.random(x.shape) < 0.1] = np.nan
    x[3] = np.nan
    x[4, 1:] = np.nan
    q = rng.random(300).astype(dtype)
    q[7], q[8], q[9] = np.nan, 0.0, 1.0
    got = hip.vecquantiles(x, q)
    assert_close(got, oracle.vecquantiles(x, q), rtol=0.0, what="vecquantiles")
    assert np.isnan(got[3]) and np.isnan(got[7]) and got[4] == x[4, 0]
    assert got[8] == np.nanmin(x[8]) and got[9] == np.nanmax(x[9])


def test_nan_mask_large(hip, train_kernel):
    """cfg5-style NaN mask at the 30-year length: 30 % all-NaN ('ocean') cells + 1 % i.i.d. NaN elsewhere, DQM
    doy/31 train + adjust: the all-NaN cells give NaN tables/scen without errors, a cell subsample equals the oracle"""
    rng = np.random.default_rng(43)
    C, T = 96, 10950
    gidx, G, W = GROUPINGS["doy31"](T)
    ref, hist, sim = _synth(rng, C, T, "+", np.float32, nan_frac=0.01)
    ocean = rng.random(C) < 0.3
    ocean[:4] = [True, False, True, False]
    for a in (ref, hist, sim):
        a[ocean] = np.nan
    q = _nodes(20)
    tnum = (np.arange(T, dtype=np.float64) + 10957) * 86400e9
    af, hq, sc = hip.dqm_train(ref, hist, gidx, G, W, q, "+")
    scen, trend = hip.dqm_adjust(sim, af, hq, sc, gidx, W, tnum, 1, "nearest", "constant", "+")
    assert np.isnan(af[ocean]).all() and np.isnan(hq[ocean]).all() and np.isnan(sc[ocean]).all() and np.isnan(scen[ocean]).all()
    assert np.isfinite(af[~ocean]).all() and np.isfinite(sc[~ocean]).all()
    sub = slice(0, 8)
    eaf, ehq, esc = oracle.dqm_train(ref[sub], hist[sub], gidx, G, W, q, "+")
    scale = float(np.nanmax(np.abs(ref)))
    assert_close(sc[sub], esc, rtol=2.4e-7, atol=2.4e-7 * scale, what="scaling")
    assert_close(hq[sub], ehq, rtol=2.4e-7, atol=2.4e-7 * scale, what="hist_q")
    eqaf, eqhq = hip.eqm_train(ref, hist, gidx, G, W, q, "+")
    oaf, ohq = oracle.eqm_train(ref[sub], hist[sub], gidx, G, W, q, "+")
    assert_close(eqhq[sub], ohq, rtol=0.0, what="eqm hist_q with NaN mask")
    assert_close(eqaf[sub], oaf, rtol=0.0, what="eqm af with NaN mask")


@pytest.mark.parametrize("method", ["eqm", "dqm"])
def test_three_and_two_stream_train_kernels_agree(hip, monkeypatch, method, train_kernel):
    """k_train_cell3 (three group streams per wave: the default for windowed day-of-year groups with <= 21 quantiles) and
    the two-stream k_train_cell (XSDBA_HIP_TRAIN3=0) select the same order statistics: bit-identical tables"""
    if train_kernel != "cell":
        pytest.skip("compares the two per-cell kernels")
    rng = np.random.default_rng(77)
    C, T, Q = 5, 10 * 365, 20
    ref = (280 + 8 * rng.standard_normal((C, T))).astype(np.float32)
    hist = (282 + 9 * rng.standard_normal((C, T))).astype(np.float32)
    hist[1, rng.random(T) < 0.05] = np.nan
    hist[2] = np.nan
    gidx = (np.arange(T) % 365).astype(np.int32)
    q = ((np.arange(Q) + 0.5) / Q).astype(np.float32)
    fn = hip.eqm_train if method == "eqm" else hip.dqm_train
    a = fn(ref, hist, gidx, 365, 31, q, "+")
    monkeypatch.setenv("XSDBA_HIP_TRAIN3", "0")
    b = fn(ref, hist, gidx, 365, 31, q, "+")
    for x, y in zip(a[:2], b[:2]):
        np.testing.assert_array_equal(x, y)
    if method == "dqm":  # the group means come from differently ordered float64 sums
        np.testing.assert_allclose(a[2], b[2], rtol=1e-6, equal_nan=True)


@pytest.mark.parametrize("method", ["eqm", "dqm"])
@pytest.mark.parametrize("T,W,Q", [(10950, 61, 20), (10950, 31, 21), (3000, 91, 7), (10950, 31, 22)])
def test_train_long_rows_and_other_quantile_counts(hip, method, T, W, Q):
    """k_train_cell3 beyond the bench shape: rows of >= 1024 values (61- / 91-day windows: the three per-lane counts no
    longer share one register), 21 nodes (every lane of a stream busy), a short series, and 22 nodes (two-stream kernel)"""
    rng = np.random.default_rng(T + W + Q)
    C = 3
    gidx = (np.arange(T) % 365).astype(np.int32)
    ref, hist, _ = _synth(rng, C, T, "+", np.float32, nan_frac=0.01)
    hist[1, :400] = np.nan
    q = _nodes(Q)
    if method == "eqm":
        af, hq = hip.eqm_train(ref, hist, gidx, 365, W, q, "+")
        eaf, ehq = oracle.eqm_train(ref, hist, gidx, 365, W, q, "+")
        assert_close(hq, ehq, rtol=0.0, what="hist_q")
        assert_close(af, eaf, rtol=0.0, what="af")
    else:
        af, hq, sc = hip.dqm_train(ref, hist, gidx, 365, W, q, "+")
        eaf, ehq, esc = oracle.dqm_train(ref, hist, gidx, 365, W, q, "+")
        assert_close(sc, esc, rtol=2e-6, what="scaling")
        assert_close(hq, ehq, rtol=3e-6, atol=3e-5, what="hist_q")


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["+", "*"])
def test_dqm_adjust_without_trend_array_is_identical(hip, kind):
    """want_trend=False: the mapping kernel evaluates the per-group regression line from the trend kernel's coefficients
    instead of reading a [cell, time] trend array — same expression, same values: scen must not change by a bit (affine and
    non-affine time axes, NaNs in sim, time steps in no group)"""
    rng = np.random.default_rng(5)
    C, T, G, Q, W = 9, 3650, 365, 20, 31
    gidx = (np.arange(T) % 365).astype(np.int32)
    gidx[rng.random(T) < 0.01] = -1
    base = 280 if kind == "+" else 5
    sim = (base + rng.gamma(2.0, 2.0, (C, T)) + 0.001 * np.arange(T)).astype(np.float32)
    sim[rng.random((C, T)) < 0.02] = np.nan
    q = ((np.arange(Q) + 0.5) / Q).astype(np.float32)
    hq = np.sort((base + rng.gamma(2.0, 2.0, (C, G, Q))).astype(np.float32), axis=-1)
    af = (rng.normal(0, 1, (C, G, Q)) if kind == "+" else 1 + 0.2 * rng.random((C, G, Q))).astype(np.float32)
    sc = (rng.normal(0, 1, (C, G)) if kind == "+" else 1 + 0.1 * rng.random((C, G))).astype(np.float32)
    for tnum in ((np.arange(T, dtype=np.float64) + 11000.0) * 86400e9, np.cumsum(1.0 + rng.random(T)) * 1e13):
        for degree in (0, 1):
            s1, tr = hip.dqm_adjust(sim, af, hq, sc, gidx, W, tnum, degree, "nearest", "constant", kind)
            s0, none = hip.dqm_adjust(sim, af, hq, sc, gidx, W, tnum, degree, "nearest", "constant", kind, want_trend=False)
            assert none is None and tr is not None
            np.testing.assert_array_equal(np.asarray(s0).view(np.uint32), np.asarray(s1).view(np.uint32))


@pytest.mark.gpu
@pytest.mark.parametrize("T,W,leap", [(10950, 31, False), (3650, 31, False), (7300, 30, False), (10958, 31, True), (10950, 1, False)])
@pytest.mark.parametrize("kind", ["+", "*"])
def test_dqm_train_group_sums_vs_running_sums(hip, T, W, leap, kind):
    """dqm_train on the three-stream kernel forms the group sums from float64 running sums over time (GroupSumHook); the
    two-stream kernel (XSDBA_HIP_TRAIN3=0) keeps sums updated by the toggles.  Same tables to 2 ulp on every series length
    class (EPT 4 / 8 / 11), even windows, leap days, NaNs, an all-NaN cell — and against the oracle."""
    import os

    rng = np.random.default_rng(T + W)
    C, Q = 6, 20
    if leap:
        g, t, y = np.empty(T, np.int32), 0, 0
        while t < T:
            n = 366 if y % 4 == 0 else 365
            m = min(n, T - t)
            g[t:t + m] = np.arange(m)
            t += m
            y += 1
        gidx, G = g, 366
    else:
        gidx, G = (np.arange(T) % 365).astype(np.int32), 365
    base = (10 + 5 * np.sin(2 * np.pi * gidx / 365.0))[None, :]
    ref = (base + rng.gamma(2.0, 1.5, (C, T))).astype(np.float32)
    hist = (base * 1.1 + rng.gamma(2.0, 1.7, (C, T))).astype(np.float32)
    hist[rng.random((C, T)) < 0.03] = np.nan
    ref[2] = np.nan
    q = ((np.arange(Q) + 0.5) / Q).astype(np.float32)
    os.environ.pop("XSDBA_HIP_TRAIN3", None)
    a3 = [np.asarray(v) for v in hip.dqm_train(ref, hist, gidx, G, W, q, kind)]
    os.environ["XSDBA_HIP_TRAIN3"] = "0"
    try:
        a2 = [np.asarray(v) for v in hip.dqm_train(ref, hist, gidx, G, W, q, kind)]
    finally:
        os.environ.pop("XSDBA_HIP_TRAIN3", None)
    exp = oracle.dqm_train(ref, hist, gidx, G, W, q, kind)
    for name, x3, x2, xe in zip(("af", "hist_q", "scaling"), a3, a2, exp):
        assert_close(x3, x2, rtol=3e-7, atol=1e-6 if kind == "+" else 0.0, what=f"{name} (three-stream vs two-stream)")
        assert_close(x3, xe, rtol=3e-7, atol=1e-6 if kind == "+" else 0.0, what=f"{name} (vs oracle)")
