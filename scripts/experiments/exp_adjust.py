import sys, time, numpy as np
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
    print(label, {k: round(v['total_ms']/v['count'], 3) for k, v in p.items() if v['count']})
tr = lambda: hip.eqm_train(ref, hist, gidx, G, W, q, '+', ctx=ctx, out=(af, hq))
ad = lambda: hip.qm_adjust(sim, af, hq, gidx, 'nearest', 'constant', '+', ctx=ctx, out=scen)
t('train', tr); t('adjust right after train', ad); t('adjust again', ad); t('adjust again', ad)
time.sleep(2.0); t('adjust after 2 s idle', ad); t('adjust again', ad)
t('train', tr); t('train again', tr); t('adjust', ad)
def both(): tr(); ad()
t('train+adjust enqueued together', both)
import os
os.environ["XSDBA_HIP_TRAIN"] = "rows"
t('train(rows)', tr); t('adjust after rows-train', ad); t('adjust again', ad)
a1 = af.download().copy(); h1 = hq.download().copy()
os.environ.pop("XSDBA_HIP_TRAIN")
t('train(cell)', tr); t('adjust after cell-train', ad)
a2 = af.download(); h2 = hq.download()
print('tables equal:', np.array_equal(a1, a2, equal_nan=True), np.array_equal(h1, h2, equal_nan=True))
