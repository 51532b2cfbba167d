#!/bin/bash
# SQ counter passes for the grouped-linear kernels (monthly table, 64 800 cells; run on the GPU box through gpurun):
#   scripts/pmc_lin2d.sh <out_tag>
TAG=${1:-r02}
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES"
P2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD"
rocprofv3 --pmc $P1 -d $REPO/gpurun_out/pmc_${TAG}_lin2d_pass1 -o p1 --output-format csv -- python3 $REPO/scripts/experiments/exp_lin2d_extrap.py > $REPO/gpurun_out/pmc_${TAG}_lin2d_pass1.log 2>&1
rocprofv3 --pmc $P2 -d $REPO/gpurun_out/pmc_${TAG}_lin2d_pass2 -o p2 --output-format csv -- python3 $REPO/scripts/experiments/exp_lin2d_extrap.py > $REPO/gpurun_out/pmc_${TAG}_lin2d_pass2.log 2>&1
cd $REPO
python3 scripts/pmc_summary.py gpurun_out/pmc_${TAG}_lin2d_pass1 gpurun_out/pmc_${TAG}_lin2d_pass2
