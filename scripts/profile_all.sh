set -e  # (run on the GPU box through gpurun: every workload of the round through scripts/profile_round.sh)
cd $GRAFT_REPO_ROOT
scripts/profile_round.sh r03_final eqm_doy31 > gpurun_out/r03/final_eqm.log 2>&1
scripts/profile_round.sh r03_finaldqm dqm_doy31 > gpurun_out/r03/final_dqm.log 2>&1
scripts/profile_round.sh r03_finalqdm qdm_doy31 > gpurun_out/r03/final_qdm.log 2>&1
scripts/profile_round.sh r03_finalqdmrw qdm_doy31_rw > gpurun_out/r03/final_qdmrw.log 2>&1
scripts/profile_round.sh r03_finalmonth eqm_month > gpurun_out/r03/final_month.log 2>&1
scripts/profile_round.sh r03_finaltime eqm_time > gpurun_out/r03/final_time.log 2>&1
EXTRA="--days 21900 --cells 32400" scripts/profile_round.sh r03_finalnan dqm_doy31_nan > gpurun_out/r03/final_nan.log 2>&1
tail -2 gpurun_out/r03/final_*.log | cut -c1-200
