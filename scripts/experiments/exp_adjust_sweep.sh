#!/bin/bash
# k_qm_adjust on 64 800 day-of-year cells: libraries (elements per round) x threads per block
cp xsdba_amd/lib/libxsdba_hip.so /tmp/libxsdba_release.so
for L in /tmp/libxsdba_release.so "$@"; do
  cp "$L" xsdba_amd/lib/libxsdba_hip.so
  for n in 384 512 640 768 1024; do
    echo "== $L threads $n: $(XSDBA_HIP_ADJ_THREADS=$n timeout -k 10 120 python3 scripts/experiments/exp_adjust_only.py 2>&1 | tail -1)"
  done
done
cp /tmp/libxsdba_release.so xsdba_amd/lib/libxsdba_hip.so
