#!/bin/bash
# The headline bench line with each of the given experiment libraries in place of the in-tree one:
#   scripts/experiments/exp_bench_libs.sh <lib1.so> ...
cp xsdba_amd/lib/libxsdba_hip.so /tmp/libxsdba_release.so
for L in /tmp/libxsdba_release.so "$@"; do
  cp "$L" xsdba_amd/lib/libxsdba_hip.so
  for i in 1 2; do
    echo "== $L: $(python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-api-level 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), d['ms_per_step'], {k:round(v['avg_ms'],3) for k,v in d['roofline']['kernels'].items() if v['avg_ms']>0.05})")"
  done
done
cp /tmp/libxsdba_release.so xsdba_amd/lib/libxsdba_hip.so
