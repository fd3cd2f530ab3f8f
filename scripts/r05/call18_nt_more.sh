#!/bin/bash
# round 5, GPU call 18: more streams as nontemporal accesses, on top of bits 21-23 (k_errlog copy-out, next-event prefetch, record image):
# bit 24 the scratch image of -k, bit 25 the staged event groups of k_chain, bit 26 the event loads of the record kernel
cd "$(dirname "$0")/../.."
O=gpurun_out/r05u; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
B=14680064
for sk in $B $((B+16777216)) $((B+33554432)) $((B+67108864)) $((B+33554432+67108864)) $((B+16777216+33554432+67108864)) $B; do echo -n "skip=$sk "; NS_DEBUG_SKIP=$sk timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-e2e 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=lambda x:round(x,2); e=d['errlog_on']; c=d['configs2']
print('step', r(d['ms_per_step']), r(d['value']/1e6), 'M; chain', r(d['kernel_ms']['k_chain']), 'k_materialise', r(d['kernel_ms']['k_materialise']), '; errlog_on', r(e['ms_per_step']), 'k_errlog', r(e['k_errlog_ms']), '; configs2', r(c['ms_per_step']), {k:r(v) for k,v in c['aligned_batch']['kernel_ms'].items() if v>0.1})"; done | tee $O/ab_nt_more.log
NS_DEBUG_SKIP=$((B+16777216+33554432+67108864)) timeout 600 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -2 | tee $O/pytest_bits.log
