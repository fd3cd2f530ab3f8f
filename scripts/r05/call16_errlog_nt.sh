#!/bin/bash
# round 5, GPU call 16: k_errlog's copy-out with nontemporal 16-byte stores (NS_DEBUG_SKIP bit 21) and with the NEXT iteration's event
# loaded in front of the copy-out instead of behind it (bit 22) — same build, same box; parity of the two forms (results are right with
# these two bits set)
cd "$(dirname "$0")/../.."
O=gpurun_out/r05s; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
for sk in 0 2097152 4194304 6291456 0 2097152 4194304 6291456; do echo -n "skip=$sk "; NS_DEBUG_SKIP=$sk timeout 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-e2e --no-configs2 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=lambda x:round(x,2); e=d['errlog_on']
print('k_errlog', r(e['k_errlog_ms']), 'ms; errlog_on', r(e['ms_per_step']), r(e['value']/1e6), 'M reads/s')"; done | tee $O/ab_errlog_nt.log
NS_DEBUG_SKIP=6291456 timeout 600 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -2 | tee $O/pytest_nt_prefetch.log
