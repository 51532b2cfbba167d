#!/bin/bash
# k_adapt_freq (selection) at several block sizes: XSDBA_HIP_AF_THREADS = 64 .. 1024, on one box
for nt in 64 128 256 512 1024; do
  echo "== threads $nt"
  XSDBA_HIP_AF_THREADS=$nt timeout -k 10 120 python3 scripts/experiments/exp_adapt_freq_select.py 2>&1 | grep "select=1" | awk 'NR%2==0'
done
