#!/bin/bash
# k_adapt_freq with each of the given libraries (phase-stop builds: -DXS_AF_STOP=1 after the rows are staged and counted, =2 without
# the tag passes), on one box:  scripts/experiments/exp_af_libs.sh <lib1.so> ...
cp xsdba_amd/lib/libxsdba_hip.so /tmp/libxsdba_release.so
for L in /tmp/libxsdba_release.so "$@"; do
  cp "$L" xsdba_amd/lib/libxsdba_hip.so
  echo "== $L"
  timeout -k 10 120 python3 scripts/experiments/exp_adapt_freq_select.py 2>&1 | grep "select=1" | awk 'NR%2==0'
done
cp /tmp/libxsdba_release.so xsdba_amd/lib/libxsdba_hip.so
