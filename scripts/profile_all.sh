set -e  # (run on the GPU box through gpurun: the workloads of the round through scripts/profile_round.sh, at the bench's default size)
# scripts/profile_all.sh <round tag> <part: a | b>   (two calls: a gpurun call ends after 20 minutes)
cd $GRAFT_REPO_ROOT
R=${1:-r04}; PART=${2:-a}; mkdir -p gpurun_out/$R
if [ "$PART" = a ]; then
scripts/profile_round.sh ${R}_final eqm_doy31 > gpurun_out/$R/final_eqm.log 2>&1
scripts/profile_round.sh ${R}_finaldqm dqm_doy31 > gpurun_out/$R/final_dqm.log 2>&1
scripts/profile_round.sh ${R}_finalqdm qdm_doy31 > gpurun_out/$R/final_qdm.log 2>&1
scripts/profile_round.sh ${R}_finalqdmrw qdm_doy31_rw > gpurun_out/$R/final_qdmrw.log 2>&1
else
scripts/profile_round.sh ${R}_finalmonth eqm_month > gpurun_out/$R/final_month.log 2>&1
scripts/profile_round.sh ${R}_finaltime eqm_time > gpurun_out/$R/final_time.log 2>&1
EXTRA="--days 21900" scripts/profile_round.sh ${R}_finalnan dqm_doy31_nan > gpurun_out/$R/final_nan.log 2>&1
fi
tail -2 gpurun_out/$R/final_*.log | cut -c1-200
