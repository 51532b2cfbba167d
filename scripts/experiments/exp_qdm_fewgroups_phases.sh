#!/bin/bash
# Phases of k_qdm_cell on un-windowed groups (64 800 cells): libraries built with -DXS_QDM_STOP=1|2|3 under scripts/ubench/build/
cp xsdba_amd/lib/libxsdba_hip.so /tmp/libxsdba_release.so
for L in scripts/ubench/build/libqdm_stop1.so scripts/ubench/build/libqdm_stop2.so scripts/ubench/build/libqdm_stop3.so /tmp/libxsdba_release.so; do
  cp "$L" xsdba_amd/lib/libxsdba_hip.so
  echo "== $L"
  timeout -k 10 120 python3 scripts/experiments/exp_qdm_fewgroups.py
done
cp /tmp/libxsdba_release.so xsdba_amd/lib/libxsdba_hip.so
