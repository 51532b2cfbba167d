#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the phase-stop instantiations of k_train_cell3 (run on the GPU box): the load-only kernel (STOP 1)
# reads a known byte count (2 series) with the kernel's own access pattern — the calibration of FETCH_SIZE for 8-byte-per-lane
# loads that MI355X_MICROARCH.md asks for before an absolute is trusted.
REPO=$(pwd); OUT=$REPO/gpurun_out/r03; mkdir -p $OUT
BIN=$REPO/scripts/ubench/build/train3_bench
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE -d $OUT/fetchcal_f -o f --output-format csv -- $BIN 64800 2 > $OUT/fetchcal_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/fetchcal_w -o w --output-format csv -- $BIN 64800 2 > $OUT/fetchcal_w.log 2>&1
cd $REPO
python3 - <<'PY'
import csv, glob, collections
for tag in ('f', 'w'):
    acc = collections.defaultdict(list)
    for f in glob.glob(f'gpurun_out/r03/fetchcal_{tag}/**/*counter_collection.csv', recursive=True):
        for row in csv.DictReader(open(f)):
            acc[row['Kernel_Name'].split('(')[0][-40:]].append(float(row['Counter_Value']))
    for k, v in acc.items():
        print(tag, k, f'{sum(v)/len(v)/1e6:.3f} M KiB = {sum(v)/len(v)*1024/1e9:.3f} GB raw')
PY
