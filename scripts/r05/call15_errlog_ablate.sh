#!/bin/bash
# round 5, GPU call 15: which phase of k_errlog costs what (NS_DEBUG_SKIP bits 16..20: no name copy / no fields / no letter columns /
# no copy-out / no position digits; the results are wrong with a bit set)
cd "$(dirname "$0")/../.."
O=gpurun_out/r05r; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
for sk in 0 65536 131072 262144 524288 1048576 196608 720896 0; do echo -n "skip=$sk "; NS_DEBUG_SKIP=$sk timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-configs2 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=lambda x:round(x,2); e=d['errlog_on']
print('k_errlog', r(e['k_errlog_ms']), 'ms; errlog_on', r(e['ms_per_step']))"; done | tee $O/ablate_errlog.log
