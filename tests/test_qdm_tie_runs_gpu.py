"""Tie runs of the QDM rank kernels (k_qdm_cell: per-word run tables, utils.rank's average ties — reference utils.py:573-638)
on series built from prescribed runs of equal values in RANK space: runs of 1, 31 .. 34, 63 .. 66 and hundreds of values that start
and end at every offset of a 32-rank word, runs that end exactly at the number of valid values, one run over the whole series,
strictly alternating pairs; with and without NaNs (the valid count is then not a multiple of 32).  Ranks and adjusted values
bit-exact against the oracle, for day-of-year windows with rank_window (the two-stream walk), a grouping that takes the
one-stream walk, time.month and one all-time group."""
import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu


def _doy_month(T):
    d0 = np.arange(T) % 365
    cum = np.concatenate([[0], np.cumsum([31, 28, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31])])
    return d0.astype(np.int32), (np.searchsorted(cum, d0, side="right") - 1).astype(np.int32)


def _runs(pattern, n, rng):
    """run lengths that add up to n"""
    if pattern == "word_edges":
        base = [1, 31, 32, 33, 34, 63, 64, 65, 66, 1, 2, 3, 95, 96, 97, 5, 127, 128, 129, 7]
    elif pattern == "long":
        base = [700, 1, 1500, 3, 64, 900, 2]
    elif pattern == "pairs":
        base = [2]
    elif pattern == "one":
        base = [n]
    elif pattern == "random":
        base = list(rng.integers(1, 80, 400))
    else:
        raise KeyError(pattern)
    out, tot, i = [], 0, 0
    off = int(rng.integers(0, 31))  # shift everything against the word grid
    if off and pattern != "one":
        out.append(off); tot += off
    while tot < n:
        r = min(int(base[i % len(base)]), n - tot)
        out.append(r); tot += r; i += 1
    return out


def _series(pattern, T, nan_frac, rng):
    valid = np.ones(T, bool)
    if nan_frac:
        valid[rng.random(T) < nan_frac] = False
    n = int(valid.sum())
    runs = _runs(pattern, n, rng)
    vals = np.repeat(np.arange(len(runs), dtype=np.float32) * np.float32(0.25) + np.float32(1.0), runs)[:n]  # sorted, one value per run
    x = np.full(T, np.nan, np.float32)
    x[np.flatnonzero(valid)] = rng.permutation(vals)   # a random time step for every rank
    return x


@pytest.mark.parametrize("pattern", ["word_edges", "long", "pairs", "one", "random"])
@pytest.mark.parametrize("gname", ["doy31_rw", "doy31_exact", "month", "time", "week40_w3"])
@pytest.mark.parametrize("T", [3650, 10950])
def test_tie_runs_vs_oracle(pattern, gname, T):
    import zlib

    from xsdba_amd import hip

    rng = np.random.default_rng(zlib.crc32(f"{pattern}{gname}{T}".encode()))
    doy, mon = _doy_month(T)
    gidx, G, W, rw = {
        "doy31_rw": (doy, 365, 31, True),
        "doy31_exact": (doy, 365, 31, False),
        "month": (mon, 12, 1, False),
        "time": (np.zeros(T, np.int32), 1, 1, False),
        "week40_w3": (((np.arange(T) // 7) % 40).astype(np.int32), 40, 3, True),   # windowed rows of ~800 steps, 40 groups: the one-stream walk
    }[gname]
    sim = np.stack([_series(pattern, T, nf, rng) for nf in (0.0, 0.013, 0.3)])
    C, Q = sim.shape[0], 15
    q = ((np.arange(Q) + 0.5) / Q).astype(np.float32)
    af = (1.0 + 0.3 * rng.standard_normal((C, G, Q))).astype(np.float32)
    r = hip.rank_groups(sim, gidx, G, W, rw)
    np.testing.assert_array_equal(np.asarray(r), oracle.rank_groups(sim, gidx, G, W, rw))
    scen, sim_q = hip.qdm_adjust(sim, af, q, gidx, W, rw, "nearest", "constant", "*")
    escen, esim_q = oracle.qdm_adjust(sim, af, q, gidx, W, rw, "nearest", "constant", "*")
    np.testing.assert_array_equal(np.asarray(sim_q), esim_q)
    np.testing.assert_array_equal(np.asarray(scen), escen)
    s2, _ = hip.qdm_adjust(sim, af, q, gidx, W, rw, "nearest", "constant", "*", want_sim_q=False)
    np.testing.assert_array_equal(np.asarray(s2), escen)
