"""Where the time of the multi-context host path goes (one GPU, two contexts): cProfile of train + adjust on a 16 384-cell host cube."""
import cProfile, os, pstats, sys, time, warnings
import numpy as np
sys.path.insert(0, ".")
import xsdba_amd as xs
from xsdba_amd import hip
from xsdba_amd.labeled import DataArray

n, T = int(sys.argv[1]) if len(sys.argv) > 1 else 16384, 10950
ctx = hip.default_context(0)
times = xs.date_range("2000-01-01", T, "noleap").values
arrs = [ctx.synth((n, T), np.float32, 0, i, 1).download() for i in range(3)]
host = [DataArray(a, ("cell", "time"), {"cell": np.arange(n), "time": times}, {"units": "K"}) for a in arrs]
grp = xs.Grouper("time.dayofyear", window=31)

def run():
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        obj = xs.EmpiricalQuantileMapping.train(host[0], host[1], group=grp, nquantiles=20, kind="+")
        return obj.adjust(host[2]).values

for spec in ("0", "0,0"):
    os.environ["XSDBA_AMD_DEVICES"] = spec
    run()
    t0 = time.perf_counter(); run(); dt = time.perf_counter() - t0
    print(f"devices={spec}: {n / dt:.0f} cells/s ({dt * 1e3:.0f} ms)")
pr = cProfile.Profile(); pr.enable(); run(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
