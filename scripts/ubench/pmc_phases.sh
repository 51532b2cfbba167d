#!/bin/bash
# Per-phase SQ counters of k_train_cell3 (run on the GPU box): the phase-stop instantiations of train3_bench are distinct kernels,
# so one rocprofv3 --pmc pass over the binary yields a counter row per phase; differences of consecutive rows attribute the
# counters to load / sort / build / toggles+recount+scan / select.   scripts/ubench/pmc_phases.sh <tag> [cells]
TAG=${1:-phases}; CELLS=${2:-64800}
REPO=$(pwd); OUT=$REPO/gpurun_out/r04; mkdir -p $OUT
BIN=$REPO/scripts/ubench/build/train3_bench
cd /tmp && export TMPDIR=/tmp
P1="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY"
P2="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_SMEM"
rocprofv3 --pmc $P1 -d $OUT/${TAG}_p1 -o p1 --output-format csv -- $BIN $CELLS 2 > $OUT/${TAG}_p1.log 2>&1
rocprofv3 --pmc $P2 -d $OUT/${TAG}_p2 -o p2 --output-format csv -- $BIN $CELLS 2 > $OUT/${TAG}_p2.log 2>&1
cd $REPO
python3 scripts/pmc_summary.py $OUT/${TAG}_p1 $OUT/${TAG}_p2 > $OUT/${TAG}_pmc.txt 2>&1
$BIN $CELLS 5 > $OUT/${TAG}_times.txt 2>&1
tail -6 $OUT/${TAG}_times.txt
