cp xsdba_amd/lib/libxsdba_hip.so /tmp/rel.so && cp build_exp/diag.so xsdba_amd/lib/libxsdba_hip.so
for p in 0 1 2 5; do XSDBA_HIP_DEBUG_PHASE=$p timeout -k 10 120 python bench.py --no-cpu-baseline --no-api-level --steps 5 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('phase', '$p', d['roofline']['kernels'])" >> gpurun_out/wb_phases2.txt || exit 1; done
cp /tmp/rel.so xsdba_amd/lib/libxsdba_hip.so
