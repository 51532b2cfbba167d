"""How many elements of the golden fractional-group cases differ between the device path and the golden vectors, and where."""
import numpy as np, sys
sys.path.insert(0, ".")
from xsdba_amd import hip
hip.default_context(0)
d = np.load("tests/golden/interp_2d_fractional.npz", allow_pickle=True)
for i in range(int(d["n"])):
    method = str(d[f"method_{i}"])
    sim, hq, af, newg, extrap = d[f"sim_{i}"], d[f"hq_{i}"], d[f"af_{i}"], d[f"newg_{i}"], str(d[f"extrap_{i}"])
    G = hq.shape[1]
    gc = np.arange(0, G + 2, dtype=np.float64)
    scen = hip.qm_adjust_linear2d(sim, af, hq, gc, newg, extrap, "+", cubic=method == "cubic")
    exp = sim + d[f"out_{i}"]
    nanmis = np.isnan(scen) != np.isnan(exp)
    both = ~np.isnan(scen) & ~np.isnan(exp)
    tol = 3e-6 if method == "linear" else 2e-5
    bad = both & ~np.isclose(scen, exp, rtol=tol, atol=tol * 10)
    print(i, method, extrap, sim.shape, "nan mismatches", int(nanmis.sum()), "value mismatches", int(bad.sum()), "max abs diff", float(np.nanmax(np.abs(scen - exp))))
    for idx in np.argwhere(bad | nanmis)[:10]:
        c, t = idx
        print("   cell", c, "t", t, "x", sim[c, t], "g", newg[t], "got", scen[c, t], "exp", exp[c, t])
