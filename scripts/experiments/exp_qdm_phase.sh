#!/bin/bash
for d in 1 2 3 0; do
  XSDBA_DBG=$d python bench.py --workload qdm_doy31 --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('dbg',$d, j['roofline']['kernels'])"
done
