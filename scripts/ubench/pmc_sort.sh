#!/bin/bash
# SQ counters of the sort's sub-phases (train3_sortstop<n> builds, STOP = 2 kernel): scripts/ubench/pmc_sort.sh
REPO=$(pwd); OUT=$REPO/gpurun_out/r04; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for n in 1 2 3 4 5 6; do
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_SALU -d $OUT/sortpmc_$n -o p --output-format csv -- $REPO/scripts/ubench/build/train3_sortstop$n 64800 2 2 > $OUT/sortpmc_$n.log 2>&1
done
cd $REPO
for n in 1 2 3 4 5 6; do echo "== sort stop $n"; python3 scripts/pmc_summary.py $OUT/sortpmc_$n; done > $OUT/sortpmc.txt
