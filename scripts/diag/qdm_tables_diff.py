import os, sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from xsdba_amd import hip
T = int(sys.argv[1]) if len(sys.argv) > 1 else 1825
d0 = np.arange(T) % 365
cum = np.concatenate([[0], np.cumsum([31, 28, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31])])
gidx = (np.searchsorted(cum, d0, side="right") - 1).astype(np.int32)
rng = np.random.default_rng(1)
C, G, Q = 3, 12, 20
sim = rng.gamma(2.0, 2.0, (C, T)).astype(np.float32)
q = ((np.arange(Q) + 0.5) / Q).astype(np.float32)
af = (1.0 + 0.3 * rng.standard_normal((C, G, Q))).astype(np.float32)
a, _ = hip.qdm_adjust(sim, af, q, gidx, 1, False, "nearest", "constant", "*", want_sim_q=False)
os.environ["XSDBA_HIP_QDM_TABLES"] = "0"
b, _ = hip.qdm_adjust(sim, af, q, gidx, 1, False, "nearest", "constant", "*", want_sim_q=False)
a, b = np.asarray(a), np.asarray(b)
bad = a != b
print("mismatch", bad.sum(), "of", bad.size)
for g in range(G):
    m = gidx == g
    print(g, m.sum() , bad[:, m].sum())
c = 0
idx = np.nonzero(bad[c])[0][:12]
for t in idx:
    g = gidx[t]
    fa, fb = a[c, t] / sim[c, t], b[c, t] / sim[c, t]
    na, nb = np.argmin(np.abs(af[c, g] - fa)), np.argmin(np.abs(af[c, g] - fb))
    ga = [(gg, int(np.argmin(np.abs(af[c, gg] - fa)))) for gg in range(G) if np.min(np.abs(af[c, gg] - fa)) < 1e-6]
    print(t, "group", g, "node on", na, "err", abs(af[c, g, na] - fa), "node off", nb, "matches", ga[:3])
