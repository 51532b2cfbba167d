#!/bin/bash
# every workload of the bench at one size, kernel level only (no CPU leg, no API leg): scripts/bench_all.sh <tag> [cells]
TAG=${1:-all}; CELLS=${2:-64800}
mkdir -p gpurun_out/r04
for w in eqm_doy31 dqm_doy31 qdm_doy31 qdm_doy31_rw eqm_month eqm_time; do
  python3 bench.py --workload $w --cells $CELLS --steps 5 --warmup 1 --no-cpu-baseline --no-api-level > gpurun_out/r04/${TAG}_$w.json 2> gpurun_out/r04/${TAG}_$w.err
done
python3 bench.py --workload dqm_doy31_nan --days 21900 --cells $((CELLS / 2)) --steps 3 --warmup 1 --no-cpu-baseline --no-api-level > gpurun_out/r04/${TAG}_dqm_doy31_nan.json 2> gpurun_out/r04/${TAG}_dqm_doy31_nan.err
python3 - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/r04/${TAG}_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        k = d["roofline"]["kernels"]
        print(f.split("${TAG}_")[1][:-5].ljust(16), "%.3f M cells/s" % (d["value"] / 1e6), "%.2f ms/step" % d["ms_per_step"], {n: v["avg_ms"] for n, v in k.items()})
    except Exception as e:
        print(f, "FAILED", e)
PY
