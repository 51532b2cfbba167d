import os, sys, numpy as np
sys.path.insert(0, '.')
from xsdba_amd import hip
ctx = hip.Context(0)
C, T, Q = 16384, 10950, 20
q = ((np.arange(Q) + 0.5) / Q).astype(np.float32)
sim = ctx.synth((C, T), np.float32, 1, 2, 3, 0)
ctx.profile_enable(True)
def t(label, f):
    ctx.profile_read(reset=True); f(); ctx.synchronize(); p = ctx.profile_read(reset=True)
    print(label, {k: round(v['total_ms']/v['count'], 3) for k, v in p.items() if v['count']}, flush=True)
d0 = np.arange(T) % 365
cum = np.concatenate([[0], np.cumsum([31, 28, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31])])
for name, gidx, G, W in (('doy31', d0.astype(np.int32), 365, 31), ('month', (np.searchsorted(cum, d0, side='right') - 1).astype(np.int32), 12, 1), ('time', np.zeros(T, np.int32), 1, 1)):
    af = ctx.empty((C, G, Q), np.float32); ctx.synchronize()
    import ctypes
    from xsdba_amd._lib import load, check
    check(load().xsdba_hip_memset(ctx.handle, ctypes.c_void_p(af.ptr), 0, ctypes.c_size_t(af.nbytes)), 'memset')
    for rw in (False, True):
        if rw and W == 1: continue
        t(f'{name} rank_window={rw} ({C} cells)', lambda: hip.qdm_adjust(sim, af, q, gidx, W, rw, 'nearest', 'constant', '+', want_sim_q=False, ctx=ctx))
    af.free()
