#!/bin/bash
# The round's remaining artefacts (run on the GPU box through gpurun, after scripts/profile_all.sh): scripts/final_extras.sh <tag>
#   bench lines of every workload WITH the committed traffic (profiles/<tag>_final*_traffic_*.json must be in the tree),
#   the BASELINE configs at their full per-GPU sizes, the per-phase counters of k_train_cell3, the sort / long / select phase
#   timings, the lane-order test of the LDS atomics, the >2^31-element check.
R=${1:-r04}; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/$R
for w in eqm_doy31 dqm_doy31 qdm_doy31 qdm_doy31_rw eqm_month eqm_time; do
  python3 bench.py --workload $w > gpurun_out/$R/line_$w.json 2> gpurun_out/$R/line_$w.err
done
python3 bench.py --workload dqm_doy31_nan --days 21900 > gpurun_out/$R/line_dqm_doy31_nan.json 2> gpurun_out/$R/line_dqm_doy31_nan.err
bash scripts/fullsize_round.sh $R > gpurun_out/$R/fullsize.txt 2>&1
bash scripts/ubench/pmc_phases.sh phases_after > gpurun_out/$R/pmc_phases.log 2>&1
(for n in 1 2 3 4 5 6; do scripts/ubench/build/train3_sortstop$n 64800 5 2 | tail -1; done) > gpurun_out/$R/sort_phases.txt 2>&1
scripts/ubench/build/long_bench 32400 3 > gpurun_out/$R/long_phases.txt 2>&1
(for n in 1 2 3 4; do scripts/ubench/build/sel_stop$n 64800 7; done; scripts/ubench/build/sel_bench 64800 7; scripts/ubench/build/sel_bench 64800 7 1) > gpurun_out/$R/sel_phases.txt 2>&1
scripts/ubench/build/atomic_order 2000 512 > gpurun_out/$R/atomic_order.txt 2>&1
python3 scripts/check_large.py 259200 > gpurun_out/$R/check_large.log 2>&1
tail -3 gpurun_out/$R/fullsize.txt gpurun_out/$R/check_large.log gpurun_out/$R/atomic_order.txt
