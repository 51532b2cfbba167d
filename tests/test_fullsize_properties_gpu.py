"""The bench workload at its FULL size (129 600 cells x 10 950 days, EQM doy/31: the per-GPU shard of north_star's 1440 x 720 cube,
what `bench.py` runs by default) checked through size-independent properties — the oracle would take minutes there:
  * position independence (a checksum of checksums in its strongest form): any block of cells regenerated on its own
    (counter-based generator, `cell_offset`) and run alone gives bit for bit the rows of the big run, at the start, in the
    middle and at the very end of the cube (32-bit index overflow would show there), and those rows equal the oracle's;
  * sortedness of the trained quantiles along the quantile axis, and their bounds;
  * idempotence: a mapping trained on identical series has all-zero additive factors and leaves any series untouched.
QDM / DQM get the same treatment on a 16 384-cell block (ranks inside [0, 1] and summing to N(N-1)/2 per group without ties)."""
from __future__ import annotations

import numpy as np
import pytest

import oracle
from xsdba_amd import hip

pytestmark = pytest.mark.gpu

T, G, W, Q = 10950, 365, 31, 20
DOY = (np.arange(T) % 365).astype(np.int32)
NODES = ((np.arange(Q) + 0.5) / Q).astype(np.float32)


@pytest.fixture(scope="module")
def ctx():
    return hip.default_context(0)


def test_eqm_doy31_full_bench_size(ctx):
    C, B = 1440 * 720 // 8, 32
    ref, hist, sim = (ctx.synth((C, T), np.float32, 0, i, 1) for i in range(3))
    af, hq = hip.eqm_train(ref, hist, DOY, G, W, NODES, "+", ctx=ctx)
    scen = hip.qm_adjust(sim, af, hq, DOY, "nearest", "constant", "+", ctx=ctx)
    ctx.synchronize()
    AF, HQ, SCEN, SIM = af.download(), hq.download(), scen.download(), sim.download()
    assert AF.shape == (C, G, Q) and SCEN.shape == (C, T)
    # sortedness and finiteness
    assert (np.diff(HQ, axis=-1) >= 0).all()
    assert np.isfinite(AF).all() and np.isfinite(SCEN).all()
    # position independence + oracle, at both ends and in the middle
    for c0 in (0, C // 2, C - B):
        r, h, s = (ctx.synth((B, T), np.float32, 0, i, 1, c0).download() for i in range(3))
        np.testing.assert_array_equal(s, SIM[c0:c0 + B])
        baf, bhq = hip.eqm_train(r, h, DOY, G, W, NODES, "+", ctx=ctx)
        bscen = hip.qm_adjust(s, baf, bhq, DOY, "nearest", "constant", "+", ctx=ctx)
        np.testing.assert_array_equal(baf, AF[c0:c0 + B])
        np.testing.assert_array_equal(bhq, HQ[c0:c0 + B])
        np.testing.assert_array_equal(bscen, SCEN[c0:c0 + B])
        eaf, ehq = oracle.eqm_train(r[:4], h[:4], DOY, G, W, NODES, "+")
        np.testing.assert_array_equal(AF[c0:c0 + 4], eaf)
        np.testing.assert_array_equal(HQ[c0:c0 + 4], ehq)
        np.testing.assert_array_equal(SCEN[c0:c0 + 4], oracle.qm_adjust(s[:4], eaf, ehq, DOY, "nearest", "constant", "+"))
    # the (time, cell) adjustment in place (k_qm_adjust_tm) on the same cube: the same values, transposed — at both ends and in the middle
    from xsdba_amd._lib import DeviceArray

    sim_tm = ctx.synth((C, T), np.float32, 0, 2, 1, time_major=True)
    scen_tm = hip.qm_adjust(sim_tm, af, hq, DOY, "nearest", "constant", "+", ctx=ctx, out_time_major=True)
    ctx.synchronize()
    for c0 in (0, C // 2, C - B):
        for t in (0, 1, T // 2, T - 1):
            row = DeviceArray.from_ptr(ctx, scen_tm.ptr + (t * C + c0) * 4, (B,), np.float32).download()
            np.testing.assert_array_equal(row, SCEN[c0:c0 + B, t])
    sim_tm.free(); scen_tm.free()
    # quantiles lie inside the range of the series they come from
    H = hist.download()
    assert (HQ.min(axis=(1, 2)) >= H.min(axis=1)).all() and (HQ.max(axis=(1, 2)) <= H.max(axis=1)).all()
    del H, AF, SCEN
    # idempotence
    af0, hq0 = hip.eqm_train(hist, hist, DOY, G, W, NODES, "+", ctx=ctx)
    scen0 = hip.qm_adjust(sim, af0, hq0, DOY, "nearest", "constant", "+", ctx=ctx)
    ctx.synchronize()
    assert not af0.download().any()
    np.testing.assert_array_equal(hq0.download(), HQ)
    np.testing.assert_array_equal(scen0.download(), SIM)
    for a in (ref, hist, sim, af, hq, scen, af0, hq0, scen0):
        a.free()


def test_qdm_and_dqm_properties_16k_cells(ctx):
    C = 16384
    ref, hist, sim = (ctx.synth((C, T), np.float32, 1, i, 3) for i in range(3))  # gamma precipitation-like, strictly positive
    af, hq = hip.eqm_train(ref, hist, DOY, G, W, NODES, "*", ctx=ctx)
    scen, sim_q = hip.qdm_adjust(sim, af, NODES, DOY, W, False, "nearest", "constant", "*", ctx=ctx)
    ctx.synchronize()
    SQ, SC = sim_q.download(), scen.download()
    assert SQ.min() >= 0.0 and SQ.max() <= 1.0 and (SC > 0).all()
    # ranks of the 30 members of a day-of-year group: (N - 1) * pct ranks sum to N (N - 1) / 2 (average ties keep the sum)
    n = T // 365
    for d in (0, 181, 364):
        np.testing.assert_allclose((SQ[:, d::365] * (n - 1)).sum(axis=1), n * (n - 1) / 2, rtol=1e-12)
    # the multiplicative mapping is monotone within a group: larger sim -> larger-or-equal percent rank
    blk = np.s_[:256, 5::365]
    order = np.argsort(sim.download()[blk], axis=1, kind="stable")
    assert (np.diff(np.take_along_axis(SQ[blk], order, axis=1), axis=1) >= 0).all()
    # DQM: the scaling of identical series is the neutral element, the trend of a detrended ... stays finite
    af2, hq2, sc2 = hip.dqm_train(hist, hist, DOY, G, W, NODES, "*", ctx=ctx)
    ctx.synchronize()
    np.testing.assert_array_equal(sc2.download(), np.ones((C, G), np.float32))
    np.testing.assert_array_equal(af2.download(), np.ones((C, G, Q), np.float32))
    for a in (ref, hist, sim, af, hq, scen, sim_q, af2, hq2, sc2):
        a.free()


def test_dqm_60_years_with_nan_mask_block_independence(ctx):
    """BASELINE configs[4]'s shape on a 4096-cell block: 21 900 days (k_train_long), 30 % all-NaN cells + 1 % i.i.d. NaN"""
    C, T2, B = 4096, 21900, 16
    doy = (np.arange(T2) % 365).astype(np.int32)
    tnum = (np.arange(T2, dtype=np.float64) + 10950.0) * 86400e9
    ref, hist, sim = (ctx.synth((C, T2), np.float32, 0, i, 4, 0, 0.30, 0.01) for i in range(3))
    af, hq, sc = hip.dqm_train(ref, hist, doy, G, W, NODES, "+", ctx=ctx)
    scen, trend = hip.dqm_adjust(sim, af, hq, sc, doy, W, tnum, 1, "nearest", "constant", "+", ctx=ctx)
    ctx.synchronize()
    AF, HQ, SC_, SCEN = af.download(), hq.download(), sc.download(), scen.download()
    dead = np.isnan(hist.download()).all(axis=1)
    assert 0.2 < dead.mean() < 0.4                      # the "ocean" cells
    assert np.isnan(AF[dead]).all() and np.isnan(SCEN[dead]).all()
    assert np.isfinite(AF[~dead]).all()
    assert (np.diff(HQ[~dead], axis=-1) >= 0).all()
    for c0 in (0, C - B):
        r, h, s = (ctx.synth((B, T2), np.float32, 0, i, 4, c0, 0.30, 0.01).download() for i in range(3))
        baf, bhq, bsc = hip.dqm_train(r, h, doy, G, W, NODES, "+", ctx=ctx)
        bscen, _ = hip.dqm_adjust(s, baf, bhq, bsc, doy, W, tnum, 1, "nearest", "constant", "+", ctx=ctx)
        np.testing.assert_array_equal(baf, AF[c0:c0 + B])
        np.testing.assert_array_equal(bsc, SC_[c0:c0 + B])
        np.testing.assert_array_equal(bscen, SCEN[c0:c0 + B])
        live = np.nonzero(~np.isnan(h).all(axis=1))[0][:2]
        eaf, ehq, esc = oracle.dqm_train(r[live], h[live], doy, G, W, NODES, "+")
        np.testing.assert_allclose(baf[live], eaf, rtol=1e-5)
        escen, _ = oracle.dqm_adjust(s[live], eaf, ehq, esc, doy, W, tnum, 1, "nearest", "constant", "+")
        np.testing.assert_allclose(bscen[live], escen, rtol=1e-5, equal_nan=True)
    for a in (ref, hist, sim, af, hq, sc, scen, trend):
        a.free()
