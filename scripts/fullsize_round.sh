#!/bin/bash
# The BASELINE configs at their full per-GPU sizes (run on the GPU box through gpurun): one bench line each into
#   gpurun_out/<tag>_fullsize_<name>.json     (copied to profiles/ afterwards)
TAG=${1:-r04}
run() { name=$1; shift; python3 bench.py "$@" --steps 3 --warmup 1 --no-api-level > gpurun_out/${TAG}_fullsize_${name}.json 2> gpurun_out/${TAG}_fullsize_${name}.err; tail -c 400 gpurun_out/${TAG}_fullsize_${name}.json | head -c 0; python3 -c "import json,sys; d=json.loads(open('gpurun_out/${TAG}_fullsize_${name}.json').read().strip().splitlines()[-1]); print('$name', d['value'], d['ms_per_step'], d['roofline']['kernels'])"; }
run cfg1_eqm_month_64800 --workload eqm_month --cells 64800
run cfg2_dqm_doy31_259200 --workload dqm_doy31 --cells 259200
run cfg3_qdm_doy31_129600 --workload qdm_doy31 --cells 129600
run cfg3_qdm_doy31_rw_129600 --workload qdm_doy31_rw --cells 129600
run cfg4_dqm_doy31_nan_21900d_129600 --workload dqm_doy31_nan --days 21900 --cells 129600
run eqm_doy31_259200 --workload eqm_doy31 --cells 259200
