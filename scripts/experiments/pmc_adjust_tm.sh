#!/bin/bash
# counters of k_qm_adjust_tm against k_qm_adjust (scripts/experiments/exp_adjust_tm.py under rocprofv3 --pmc)
REPO=$(pwd); OUT=$REPO/gpurun_out/r04; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES -d $OUT/tm_p1 -o p --output-format csv -- python3 $REPO/scripts/experiments/exp_adjust_tm.py > $OUT/tm_p1.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/tm_p2 -o p --output-format csv -- python3 $REPO/scripts/experiments/exp_adjust_tm.py > $OUT/tm_p2.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/tm_p3 -o p --output-format csv -- python3 $REPO/scripts/experiments/exp_adjust_tm.py > $OUT/tm_p3.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY -d $OUT/tm_p4 -o p --output-format csv -- python3 $REPO/scripts/experiments/exp_adjust_tm.py > $OUT/tm_p4.log 2>&1
cd $REPO
python3 scripts/pmc_summary.py $OUT/tm_p1 $OUT/tm_p2 $OUT/tm_p3 $OUT/tm_p4 2>&1 | grep -A9 "k_qm_adjust" 
