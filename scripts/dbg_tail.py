import traceback, warnings, numpy as np, sys
sys.path.insert(0, ".")
import xsdba_amd as xs
from xsdba_amd.labeled import DataArray
rng = np.random.default_rng(0)
def series(start, n, hi, first=None):
    vals = rng.uniform(0.001, hi, size=(n, 3))
    if first is not None: vals[0] = first
    return DataArray(vals, ("time", "lat"), {"time": xs.date_range(start, n, "standard").values, "lat": np.arange(3)}, {"units": "mm d-1"})
ref, hist, sim = series("1990-01-01", 4000, 20), series("1990-01-01", 4000, 10), series("2021-01-01", 3000, 10, 300)
warnings.simplefilter("ignore")
try:
    adj = xs.DetrendedQuantileMapping.train(ref, hist, kind="*", group="time", max_tail_factor=10)
    print(adj.adjust(sim).values[0])
except Exception as e:
    traceback.print_exc()
    c = e.__cause__
    while c is not None:
        print("CAUSE:", repr(c)); traceback.print_tb(c.__traceback__); c = c.__cause__
