#!/bin/bash
# k_qdm_cell time + FETCH/WRITE per launch: /tmp/qdm_traffic.sh <tag>
TAG=$1; REPO=$(pwd); OUT=$REPO/gpurun_out/r04; mkdir -p $OUT
python3 bench.py --workload qdm_doy31_rw --cells 64800 --steps 5 --warmup 1 --no-cpu-baseline --no-api-level 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['kernels'])"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE -d $OUT/${TAG}_fetch -o f --output-format csv -- python3 $REPO/bench.py --workload qdm_doy31_rw --cells 64800 --steps 2 --warmup 1 --no-cpu-baseline --no-api-level > $OUT/${TAG}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/${TAG}_write -o w --output-format csv -- python3 $REPO/bench.py --workload qdm_doy31_rw --cells 64800 --steps 2 --warmup 1 --no-cpu-baseline --no-api-level > $OUT/${TAG}_write.log 2>&1
cd $REPO
python3 - <<PY
import csv, glob, collections
for kind in ("fetch", "write"):
    f = glob.glob("$OUT/${TAG}_%s/**/*counter_collection.csv" % kind, recursive=True)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:40]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        if "qdm_cell" in k or "train_cell3" in k: print(kind, k, "KiB/launch %.1f" % (sum(v) / len(v)), "GB %.3f" % (sum(v) / len(v) * 1024 / 1e9 * (2 if kind == "fetch" else 1)))
PY
