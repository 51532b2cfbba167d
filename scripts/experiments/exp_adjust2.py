import os, sys, time, numpy as np
sys.path.insert(0, '.')
from xsdba_amd import hip
ctx = hip.Context(0)
C, T, Q, G, W = 64800, 10950, 20, 365, 31
gidx = (np.arange(T) % 365).astype(np.int32)
q = ((np.arange(Q) + 0.5) / Q).astype(np.float32)
ref, hist, sim = [ctx.synth((C, T), np.float32, 0, i, 1, 0) for i in range(3)]
af, hq, scen = ctx.empty((C, G, Q), np.float32), ctx.empty((C, G, Q), np.float32), ctx.empty((C, T), np.float32)
ctx.synchronize()
ctx.profile_enable(True)
def t(label, f):
    ctx.profile_read(reset=True); f(); ctx.synchronize(); p = ctx.profile_read(reset=True)
    print(label, {k: round(v['total_ms']/v['count'], 3) for k, v in p.items() if v['count']}, flush=True)
tr = lambda: hip.eqm_train(ref, hist, gidx, G, W, q, '+', ctx=ctx, out=(af, hq))
ad = lambda: hip.qm_adjust(sim, af, hq, gidx, 'nearest', 'constant', '+', ctx=ctx, out=scen)
os.environ["XSDBA_HIP_TRAIN"] = "rows"
t('train(rows)', tr); t('adjust (no cell kernel ran yet)', ad); t('adjust again', ad)
# month grouping through the cell kernel on a SMALL problem (different plan) to see whether merely running it flips the state
os.environ.pop("XSDBA_HIP_TRAIN")
small = [ctx.synth((256, T), np.float32, 0, i, 1, 0) for i in range(2)]
t('cell kernel on 256 cells', lambda: hip.eqm_train(small[0], small[1], gidx, G, W, q, '+', ctx=ctx))
t('adjust after small cell kernel', ad); t('adjust again', ad)
t('train(cell) full', tr); t('adjust', ad)
print('--- part 2: rows tables in (af,hq); cell kernel writes OTHER buffers', flush=True)
os.environ["XSDBA_HIP_TRAIN"] = "rows"
t('train(rows)', tr); t('adjust', ad)
os.environ.pop("XSDBA_HIP_TRAIN")
af2, hq2 = ctx.empty((C, G, Q), np.float32), ctx.empty((C, G, Q), np.float32)
t('train(cell) into other buffers', lambda: hip.eqm_train(ref, hist, gidx, G, W, q, '+', ctx=ctx, out=(af2, hq2)))
t('adjust with rows tables', ad)
ad2 = lambda: hip.qm_adjust(sim, af2, hq2, gidx, 'nearest', 'constant', '+', ctx=ctx, out=scen)
t('adjust with cell tables', ad2); t('adjust with rows tables', ad); t('adjust with cell tables', ad2)
a, b = af.download(), af2.download()
print('bitwise equal af:', np.array_equal(a.view(np.uint32), b.view(np.uint32)), ' hq:', np.array_equal(hq.download().view(np.uint32), hq2.download().view(np.uint32)))
d = a.view(np.uint32) != b.view(np.uint32)
print('n diff', d.sum(), a[d][:5], b[d][:5])
