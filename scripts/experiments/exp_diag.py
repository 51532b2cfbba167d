import os, sys, numpy as np
sys.path.insert(0, '.')
os.environ["XSDBA_HIP_DEBUG_PHASE"] = "9"
from xsdba_amd import hip
ctx = hip.Context(0)
C, T, Q, G, W = 4096, 10950, 20, 365, 31
gidx = (np.arange(T) % 365).astype(np.int32)
q = ((np.arange(Q) + 0.5) / Q).astype(np.float32)
ref, hist = [ctx.synth((C, T), np.float32, 0, i, 1, 0) for i in range(2)]
af, hq = hip.eqm_train(ref, hist, gidx, G, W, q, '+', ctx=ctx)
ctx.synchronize()
v = hq.download().reshape(-1)[:8]
names = ['toggles', 'popcount+scan', 'mean+index math', 'select(lo)', 'value lookup', 'next bit(+2nd select)', 'hi value+lerp+stores', 'initial builds']
tot = v[:8].sum()
for n, x in zip(names, v):
    print(f'{n:26s} {x:12.1f} kcycles  {100*x/tot:5.1f} %')
print('cells per block:', C / min(C, 1024), ' (wave 0 of block 0, both arrays)')
