#!/bin/bash
# One call on the GPU box (through gpurun) that produces every artefact copied into profiles/:
#   scripts/profile_round.sh <tag> [workload]
#   1. python3 bench.py                                   -> gpurun_out/<tag>_bench.json      (the judged JSON line)
#   2. rocprofv3 --kernel-trace --stats -- python3 bench.py -> gpurun_out/<tag>_kernel_stats.csv
#   3. rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, no tracing)  -> <tag>_pmc_{fetch,write}_size.csv
#   4. rocprofv3 --pmc SQ_* (two passes)                  -> <tag>_pmc_sq.txt  (VALU / LDS / wait shares)
# The profiled passes run without the API leg (--no-api-level): its 16 384-cell host sample would mix short launches into the
# per-kernel averages.  STATS_ONLY=1: step 2 alone.
# then: python3 scripts/profile_collect.py <tag> [workload] here (CPU) folds them into profiles/.
TAG=${1:-r01}; W=${2:-eqm_doy31}; X=${EXTRA:-}   # EXTRA: further bench.py arguments (e.g. "--days 21900 --cells 32400")
REPO=$(pwd); OUT=$REPO/gpurun_out
mkdir -p $OUT
if [ -z "$STATS_ONLY" ]; then python3 bench.py --workload $W $X > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err; fi
cd /tmp && export TMPDIR=/tmp
# (--no-api-level: the API leg also runs a 16 384-cell host sample, whose short launches would pull the kernel averages down)
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_stats -o s --output-format csv -- python3 $REPO/bench.py --workload $W $X --steps 3 --warmup 1 --no-cpu-baseline --no-api-level > $OUT/${TAG}_stats.log 2>&1
if [ -n "$STATS_ONLY" ]; then cd $REPO; find $OUT/${TAG}_stats -name "*kernel_stats.csv" | head -2; exit 0; fi
rocprofv3 --pmc FETCH_SIZE -d $OUT/${TAG}_fetch -o f --output-format csv -- python3 $REPO/bench.py --workload $W $X --steps 2 --warmup 1 --no-cpu-baseline --no-api-level > $OUT/${TAG}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/${TAG}_write -o w --output-format csv -- python3 $REPO/bench.py --workload $W $X --steps 2 --warmup 1 --no-cpu-baseline --no-api-level > $OUT/${TAG}_write.log 2>&1
P1="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES"
P2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD"
rocprofv3 --pmc $P1 -d $OUT/${TAG}_sq1 -o p1 --output-format csv -- python3 $REPO/bench.py --workload $W $X --steps 2 --warmup 1 --no-cpu-baseline --no-api-level > $OUT/${TAG}_sq1.log 2>&1
rocprofv3 --pmc $P2 -d $OUT/${TAG}_sq2 -o p2 --output-format csv -- python3 $REPO/bench.py --workload $W $X --steps 2 --warmup 1 --no-cpu-baseline --no-api-level > $OUT/${TAG}_sq2.log 2>&1
cd $REPO
python3 scripts/pmc_summary.py $OUT/${TAG}_sq1 $OUT/${TAG}_sq2 > $OUT/${TAG}_pmc_sq.txt 2>&1
cat $OUT/${TAG}_bench.json | cut -c1-300
find $OUT/${TAG}_stats -name "*kernel_stats.csv" | head -2
