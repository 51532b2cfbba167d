import sys, time
sys.path.insert(0, ".")
from xsdba_amd._lib import Context
from xsdba_amd import hip
c0 = hip.default_context(0)
import os
for mode in ("probe", "assume"):
    os.environ["XSDBA_HIP_ATOMIC_ORDER"] = mode
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); c = Context(0); ts.append(time.perf_counter() - t0); assert c.lane_ordered_atomics; c.close()
    print(mode, "context creation ms:", [round(t * 1e3, 2) for t in ts])
