"""PCIe-inclusive throughput of EQM doy/31 train+adjust when the cubes live in host memory (never bench.py's `value`):
(a) one in-core call per stage on pageable host arrays, (b) the streamed pipeline of xsdba_amd/streaming.py."""
import json, sys, time
import numpy as np
sys.path.insert(0, ".")
from xsdba_amd import hip
from xsdba_amd.streaming import train_adjust_streamed

C, T = int(sys.argv[1]) if len(sys.argv) > 1 else 64800, 10950
G, W, Q = 365, 31, 20
ctx = hip.default_context(0)
doy = (np.arange(T) % 365).astype(np.int32)
q = ((np.arange(Q) + 0.5) / Q).astype(np.float32)
# device-generated, then downloaded: the same synthetic cubes as bench.py
ref, hist, sim = (ctx.synth((C, T), np.float32, 0, i, 1, 0).download() for i in range(3))
out = {"cells": C, "days": T}
t0 = time.perf_counter()
af, hq = hip.eqm_train(ref, hist, doy, G, W, q, "+")
scen0 = hip.qm_adjust(sim, af, hq, doy, "nearest", "constant", "+")
dt = time.perf_counter() - t0
out["in_core_pageable"] = {"seconds": round(dt, 4), "cells_per_s": round(C / dt, 1)}
for B in (4096, 8192, 16384):
    st = {}
    scen = np.empty_like(sim)
    train_adjust_streamed("eqm", ref, hist, sim, doy, G, W, q, "+", block_cells=B, scen=scen, stats=st)
    assert np.array_equal(scen, scen0, equal_nan=True)
    out[f"streamed_B{B}"] = {k: (round(v, 4) if isinstance(v, float) else v) for k, v in st.items()}
print(json.dumps(out))
