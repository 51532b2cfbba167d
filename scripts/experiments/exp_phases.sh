#!/bin/bash
# Phase timing of a kernel from experiment builds (run on the GPU box through gpurun):
#   scripts/experiments/exp_phases.sh <workload> <lib1.so> <lib2.so> ...   (libraries under build_exp/, built with -DXS_*_STOP=n)
# Each library is copied over the in-tree one (on the box's scratch copy only) and the bench line's kernel table printed.
W=$1; shift
cp xsdba_amd/lib/libxsdba_hip.so /tmp/libxsdba_release.so
for L in "$@"; do
  cp "$L" xsdba_amd/lib/libxsdba_hip.so
  echo "== $L"
  timeout -k 10 120 python3 bench.py --workload $W --steps 5 --warmup 1 --no-cpu-baseline | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['roofline']['kernels'])"
done
cp /tmp/libxsdba_release.so xsdba_amd/lib/libxsdba_hip.so
