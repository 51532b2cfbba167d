"""Where the host time of the drop-in API goes (device-resident EQM doy/31): cProfile of train + adjust."""
import cProfile, pstats, sys, time, numpy as np
sys.path.insert(0, ".")
import xsdba_amd as xs
from xsdba_amd import hip
from xsdba_amd.labeled import DataArray
ctx = hip.default_context(0)
C, T = 64800, 10950
cubes = [ctx.synth((C, T), np.float32, 0, i, 3, 0) for i in range(3)]
time_c = xs.date_range("2000-01-01", T, "noleap").values
mk = lambda d: DataArray(d, ("cell", "time"), {"time": time_c, "cell": np.arange(C)}, {"units": "K"})
ref, hist, sim = map(mk, cubes)
grp = xs.Grouper("time.dayofyear", window=31)
def step():
    obj = xs.EmpiricalQuantileMapping.train(ref, hist, nquantiles=20, group=grp, kind="+")
    scen = obj.adjust(sim)
    ctx.synchronize()
    return scen
step(); step()
t0 = time.perf_counter(); step(); print("step", time.perf_counter() - t0)
pr = cProfile.Profile(); pr.enable(); step(); step(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
