"""Layout kernels (k_transpose_tile; k_transpose_dense from 64 x 64 up): time-major cubes in and out through the C ABI, aligned and ragged tile edges.  With zero
factors the adjustment is the identity, so the output must equal the input bit for bit in either layout."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("C,T", [(128, 1024), (64, 10950), (192, 366), (66, 130), (65, 129), (4, 64), (260, 3650), (1000, 730), (300, 257), (129, 255), (70, 10950)])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_time_major_round_trip(C, T, dtype):
    from xsdba_amd import hip

    rng = np.random.default_rng(C * 100003 + T)
    sim = rng.normal(280, 10, (C, T)).astype(dtype)
    sim[rng.random((C, T)) < 0.01] = np.nan
    Q = 5
    hq = np.sort(rng.normal(280, 10, (C, 1, Q)).astype(dtype), axis=-1)
    af = np.zeros((C, 1, Q), dtype)
    gidx = np.zeros(T, np.int32)
    tm = np.ascontiguousarray(sim.T)  # (time, cell) storage
    # time-major in, cell-major out
    out = np.asarray(hip.qm_adjust(tm.T, af, hq, gidx, "nearest", "constant", "+"))
    np.testing.assert_array_equal(out, sim)
    # cell-major in, time-major out
    out_tm = np.asarray(hip.qm_adjust(sim, af, hq, gidx, "nearest", "constant", "+", out_time_major=True))
    np.testing.assert_array_equal(out_tm, sim)
    # both
    out_tm2 = np.asarray(hip.qm_adjust(tm.T, af, hq, gidx, "nearest", "constant", "+", out_time_major=True))
    np.testing.assert_array_equal(out_tm2, sim)
