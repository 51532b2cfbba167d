"""xsdba_hip_malloc / xsdba_hip_free keep freed buffers of >= 1 MiB for the next request of the same size (include/xsdba_hip.h):
reuse is stream-ordered, so results of a loop that frees and re-allocates its outputs every pass are those of the first pass;
XSDBA_HIP_POOL_MB=0 turns the pool off."""
from __future__ import annotations

import numpy as np
import pytest

from xsdba_amd import hip

pytestmark = pytest.mark.gpu


def test_freed_buffer_serves_the_next_request_of_its_size(monkeypatch):
    monkeypatch.delenv("XSDBA_HIP_POOL_MB", raising=False)
    ctx = hip.Context(0)
    a = ctx.empty((1 << 19,), np.float32)  # 2 MiB
    pa = a.ptr
    a.free()
    b = ctx.empty((1 << 18,), np.float64)  # same bytes
    assert b.ptr == pa
    c = ctx.empty((1 << 19,), np.float32)  # b is in use: a fresh buffer
    assert c.ptr != pa
    b.free()
    c.free()
    ctx.close()


def test_pool_off(monkeypatch):
    monkeypatch.setenv("XSDBA_HIP_POOL_MB", "0")
    ctx = hip.Context(0)
    a = ctx.empty((1 << 19,), np.float32)
    a.upload(np.arange(1 << 19, dtype=np.float32))
    a.free()
    b = ctx.empty((1 << 19,), np.float32)
    b.upload(np.ones(1 << 19, np.float32))
    assert float(b.download().sum()) == float(1 << 19)
    b.free()
    ctx.close()


def test_loop_that_reallocates_its_outputs_gives_the_first_pass(monkeypatch):
    monkeypatch.delenv("XSDBA_HIP_POOL_MB", raising=False)
    ctx = hip.Context(0)
    rng = np.random.default_rng(5)
    C, T, Q = 96, 3650, 20
    doy = (np.arange(T) % 365).astype(np.int32)
    q = ((np.arange(Q) + 0.5) / Q)
    ref, hist, sim = (ctx.to_device((rng.random((C, T)) * 5 + i).astype(np.float32)) for i in range(3))
    first = None
    for _ in range(4):
        af, hq = hip.eqm_train(ref, hist, doy, 365, 31, q, "+", ctx=ctx)
        scen = hip.qm_adjust(sim, af, hq, doy, "nearest", "constant", "+", ctx=ctx)
        out = (af.download(), hq.download(), scen.download())
        if first is None:
            first = out
        for x, y in zip(first, out):
            np.testing.assert_array_equal(x, y)
        for b in (af, hq, scen):
            b.free()
    ctx.close()
