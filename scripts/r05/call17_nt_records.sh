#!/bin/bash
# round 5, GPU call 17: nontemporal 16-byte stores for the record image too (NS_DEBUG_SKIP bit 23: record kernel, k_qualities, dense
# kernel; bits 21 + 22: k_errlog's copy-out nontemporal + next event loaded in front of it) — parity with all three set, then the step,
# errlog_on and configs2 with and without
cd "$(dirname "$0")/../.."
O=gpurun_out/r05t; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
NS_DEBUG_SKIP=14680064 timeout 600 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -2 | tee $O/pytest_bits.log
for sk in 6291456 14680064 6291456 14680064; do echo -n "skip=$sk "; NS_DEBUG_SKIP=$sk timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-e2e 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=lambda x:round(x,2); e=d['errlog_on']; c=d['configs2']
print('step', r(d['ms_per_step']), r(d['value']/1e6), 'M; k_materialise', r(d['kernel_ms']['k_materialise']), 'unaligned rec', r(d['unaligned_batch']['kernel_ms']['k_materialise']), '; serial', r(d['serial']['ms_per_step']), '; errlog_on', r(e['ms_per_step']), 'k_errlog', r(e['k_errlog_ms']), '; configs2', r(c['ms_per_step']), {k:r(v) for k,v in c['aligned_batch']['kernel_ms'].items() if v>0.01})"; done | tee $O/ab_nt_records.log
