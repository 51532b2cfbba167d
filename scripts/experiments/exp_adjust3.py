import os, sys, time, subprocess, numpy as np
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
def smi(tag):
    try:
        o = subprocess.run(['rocm-smi', '--showclocks', '--showpower', '--showtemp'], capture_output=True, text=True, timeout=20).stdout
        keep = [l.strip() for l in o.splitlines() if any(k in l for k in ('sclk', 'mclk', 'Power', 'Temperature (Sensor junction)', 'fclk'))]
        print(tag, ' | '.join(keep[:8]), flush=True)
    except Exception as e:
        print('smi failed', e)
def t(label, f):
    ctx.profile_read(reset=True); f(); ctx.synchronize(); p = ctx.profile_read(reset=True)
    print(label, {k: round(v['total_ms']/v['count'], 3) for k, v in p.items() if v['count']}, flush=True)
tr = lambda: hip.eqm_train(ref, hist, gidx, G, W, q, '+', ctx=ctx, out=(af, hq))
ad = lambda: hip.qm_adjust(sim, af, hq, gidx, 'nearest', 'constant', '+', ctx=ctx, out=scen)
os.environ["XSDBA_HIP_TRAIN"] = "rows"
t('train(rows)', tr); t('adjust (fresh)', ad)
smi('before cell:')
os.environ.pop("XSDBA_HIP_TRAIN")
t('train(cell)', tr); t('adjust', ad)
smi('after cell:')
for s in (5, 15, 30):
    time.sleep(s); t(f'adjust after +{s}s idle', ad)
smi('end:')
