#!/bin/bash
# SQ counter passes for the dominant kernels of one bench workload (run on the GPU box through gpurun):
#   scripts/pmc_sq.sh <workload> <out_tag>
# Writes gpurun_out/pmc_<tag>_pass{1,2}/..._counter_collection.csv ; summarise with scripts/pmc_summary.py
W=${1:-eqm_doy31}; TAG=${2:-r01}
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES"
P2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD"
rocprofv3 --pmc $P1 -d $REPO/gpurun_out/pmc_${TAG}_pass1 -o p1 --output-format csv -- python3 $REPO/bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline > $REPO/gpurun_out/pmc_${TAG}_pass1.log 2>&1
rocprofv3 --pmc $P2 -d $REPO/gpurun_out/pmc_${TAG}_pass2 -o p2 --output-format csv -- python3 $REPO/bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline > $REPO/gpurun_out/pmc_${TAG}_pass2.log 2>&1
cd $REPO
python3 scripts/pmc_summary.py gpurun_out/pmc_${TAG}_pass1 gpurun_out/pmc_${TAG}_pass2
