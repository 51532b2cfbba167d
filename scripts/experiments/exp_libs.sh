#!/bin/bash
# Run a script with each of the given experiment libraries in place of the in-tree one (on the GPU box's scratch copy):
#   scripts/experiments/exp_libs.sh <script.py> <lib1.so> <lib2.so> ...
S=$1; shift
cp xsdba_amd/lib/libxsdba_hip.so /tmp/libxsdba_release.so
for L in "$@" /tmp/libxsdba_release.so; do
  cp "$L" xsdba_amd/lib/libxsdba_hip.so
  echo "== $L"
  timeout -k 10 120 python3 $S
done
cp /tmp/libxsdba_release.so xsdba_amd/lib/libxsdba_hip.so
