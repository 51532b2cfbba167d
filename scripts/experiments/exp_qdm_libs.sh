#!/bin/bash
# k_qdm_cell (doy/31 rank_window, time.month, time; 64 800 cells) with each of the given libraries, on one box:
#   scripts/experiments/exp_qdm_libs.sh <lib1.so> ...
cp xsdba_amd/lib/libxsdba_hip.so /tmp/libxsdba_release.so
for L in /tmp/libxsdba_release.so "$@" /tmp/libxsdba_release.so; do
  cp "$L" xsdba_amd/lib/libxsdba_hip.so
  echo "== $L: $(timeout -k 10 120 python3 scripts/experiments/exp_qdm_rw.py 2>&1 | tail -1) $(timeout -k 10 120 python3 scripts/experiments/exp_qdm_fewgroups.py 2>&1 | tail -2 | tr '\n' ' ')"
done
cp /tmp/libxsdba_release.so xsdba_amd/lib/libxsdba_hip.so
