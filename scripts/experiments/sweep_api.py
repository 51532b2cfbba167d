"""The drop-in classes on device-resident cubes over groupers / methods: wall time per train + adjust (host overheads included)."""
import sys, time, numpy as np
sys.path.insert(0, ".")
import xsdba_amd as xs
from xsdba_amd import hip
from xsdba_amd.labeled import DataArray
ctx = hip.default_context(0)
C, T = int(sys.argv[1]) if len(sys.argv) > 1 else 16200, 10950
time_c = xs.date_range("2000-01-01", T, "noleap").values
def cubes(gen, units):
    mk = lambda d: DataArray(d, ("cell", "time"), {"time": time_c, "cell": np.arange(C)}, {"units": units})
    return [mk(ctx.synth((C, T), np.float32, gen, i, 3, 0)) for i in range(3)]
def timed(label, fn):
    fn(); ctx.synchronize()
    t0 = time.perf_counter(); fn(); ctx.synchronize(); dt = time.perf_counter() - t0
    print(f"{label:58s} {dt*1e3:8.2f} ms  {C/dt/1e6:6.2f} M cells/s", flush=True)
for gen, kind, units in ((0, "+", "K"), (1, "*", "mm/d")):
    ref, hist, sim = cubes(gen, units)
    for gname, win in (("time", 1), ("time.month", 1), ("time.season", 1), ("time.dayofyear", 31), ("time.dayofyear", 1)):
        grp = xs.Grouper(gname, window=win) if gname != "time" else "time"
        for cls in (xs.EmpiricalQuantileMapping, xs.DetrendedQuantileMapping, xs.QuantileDeltaMapping):
            def step():
                obj = cls.train(ref, hist, nquantiles=20, group=grp, kind=kind)
                return obj.adjust(sim)
            try:
                timed(f"{cls.__name__[:3]} {gname} w={win} {kind}", step)
            except Exception as e:  # noqa: BLE001
                print(f"{cls.__name__[:3]} {gname} w={win} {kind}: {type(e).__name__}: {str(e)[:100]}")
