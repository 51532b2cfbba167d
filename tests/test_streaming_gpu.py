"""Out-of-core driver (xsdba_amd/streaming.py): blocks of cells streamed through three HIP streams give
bit-identical results to one in-core call, for every method, with a block size that does not divide the cell count."""
from __future__ import annotations

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _data(C, T, kind, seed=0):
    rng = np.random.default_rng(seed)
    doy = (np.arange(T) % 365).astype(np.int32)
    if kind == "+":
        base = 280 - 20 * np.cos(2 * np.pi * (doy + 1) / 365)
        mk = lambda a, b: (base * a + 2 * rng.random((C, T)) + b + rng.standard_normal((C, 1))).astype(np.float32)  # noqa: E731
        return mk(1.0, 0.0), mk(0.95, 0.5), mk(0.95, 1.5), doy
    g = lambda k, th: rng.gamma(k, th, (C, T)).astype(np.float32)  # noqa: E731
    return g(4, 7.5), g(8.15, 3.68), g(16, 2.63), doy


@pytest.mark.parametrize("method,kind", [("eqm", "+"), ("dqm", "+"), ("qdm", "*")])
def test_streamed_equals_in_core(method, kind):
    from xsdba_amd import hip
    from xsdba_amd.streaming import train_adjust_streamed

    hip.default_context(0)
    C, T, G, W = 700, 1460, 365, 31
    ref, hist, sim, doy = _data(C, T, kind)
    sim[5, 100:130] = np.nan
    q = ((np.arange(15) + 0.5) / 15).astype(np.float32)
    tnum = (np.arange(T, dtype=np.float64) + 11000.0) * 86400e9
    stats = {}
    scen, af, hq = train_adjust_streamed(method, ref, hist, sim, doy, G, W, q, kind, block_cells=256, time_num=tnum,
                                         keep_tables=True, stats=stats)
    assert stats["blocks"] == 3 and stats["cells"] == C
    if method == "dqm":
        eaf, ehq, esc = hip.dqm_train(ref, hist, doy, G, W, q, kind)
        escen, _ = hip.dqm_adjust(sim, eaf, ehq, esc, doy, W, tnum, 1, "nearest", "constant", kind)
    else:
        eaf, ehq = hip.eqm_train(ref, hist, doy, G, W, q, kind)
        if method == "qdm":
            escen, _ = hip.qdm_adjust(sim, eaf, q, doy, W, False, "nearest", "constant", kind)
        else:
            escen = hip.qm_adjust(sim, eaf, ehq, doy, "nearest", "constant", kind)
    assert np.array_equal(af, eaf, equal_nan=True) and np.array_equal(hq, ehq, equal_nan=True)
    assert np.array_equal(scen, escen, equal_nan=True)
    # ... and the streamed result itself is held to the ORACLE (not only to the one-piece device call) on cells at both ends of
    # the cube and on both sides of a block boundary
    import oracle
    from conftest import assert_close, assert_close_or_tie, assert_dqm_close_or_flip

    sel = [0, 5, 255, 256, 511, 512, C - 1]
    if method == "dqm":
        oaf, ohq, osc = oracle.dqm_train(ref[sel], hist[sel], doy, G, W, q, kind)
        assert_close(af[sel], oaf, rtol=3e-7, atol=3e-5, what="streamed dqm af vs oracle")
        assert_close(hq[sel], ohq, rtol=3e-7, atol=3e-5, what="streamed dqm hist_q vs oracle")
        oscen, otrend = oracle.dqm_adjust(sim[sel], oaf, ohq, osc, doy, W, tnum, 1, "nearest", "constant", kind)
        assert_dqm_close_or_flip(scen[sel], oscen, sim[sel], osc, otrend, ohq, oaf, doy, kind, rtol=1e-5, what="streamed dqm scen vs oracle")
    else:
        oaf, ohq = oracle.eqm_train(ref[sel], hist[sel], doy, G, W, q, kind)
        assert np.array_equal(af[sel], oaf, equal_nan=True) and np.array_equal(hq[sel], ohq, equal_nan=True)
        if method == "qdm":
            oscen, osq = oracle.qdm_adjust(sim[sel], oaf, q, doy, W, False, "nearest", "constant", kind)
            assert_close_or_tie(scen[sel], oscen, sim[sel], q, oaf, doy, kind, rtol=1e-6, what="streamed qdm scen vs oracle", newx=osq, max_frac=5e-3)
        else:
            oscen = oracle.qm_adjust(sim[sel], oaf, ohq, doy, "nearest", "constant", kind)
            assert_close_or_tie(scen[sel], oscen, sim[sel], ohq, oaf, doy, kind, rtol=1e-6, what="streamed eqm scen vs oracle")
    # a caller-provided output cube is filled in place; one block covering everything also works
    out = np.full((C, T), -1.0, np.float32)
    r = train_adjust_streamed(method, ref, hist, sim, doy, G, W, q, kind, block_cells=4096, time_num=tnum, scen=out)
    assert r is out and np.array_equal(out, escen, equal_nan=True)


def test_streamed_argument_checks():
    from xsdba_amd.streaming import train_adjust_streamed

    x = np.zeros((4, 400), np.float32)
    q = np.array([0.25, 0.75], np.float32)
    with pytest.raises(ValueError, match="method"):
        train_adjust_streamed("loci", x, x, x, None, 1, 1, q)
    with pytest.raises(ValueError, match="C-contiguous float32"):
        train_adjust_streamed("eqm", x.astype(np.float64), x, x, None, 1, 1, q)
    with pytest.raises(ValueError, match="time_num"):
        train_adjust_streamed("dqm", x, x, x, None, 1, 1, q)
