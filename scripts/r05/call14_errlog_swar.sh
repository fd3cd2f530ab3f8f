#!/bin/bash
# round 5, GPU call 14: k_errlog with its two letter columns as straight-line byte-permute code (<= 16 letters, plain bases) instead of the
# divergent per-byte loop: parity (error profile bytes GPU == oracle in every parity case), then the errlog_on leg
cd "$(dirname "$0")/../.."
O=gpurun_out/r05q; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cli.py tests/test_gpu_metagenome.py tests/test_gpu_transcriptome.py -x -q 2>&1 | tail -3 | tee $O/pytest.log
for i in 1 2; do timeout 240 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-e2e --no-configs2 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=lambda x:round(x,2); e=d['errlog_on']
print('step', r(d['ms_per_step']), 'errlog_on', r(e['ms_per_step']), 'ms', r(e['value']/1e6), 'M reads/s; k_errlog', r(e['k_errlog_ms']), 'ms', r(e['k_errlog_store_gb_per_s']), 'GB/s')"; done | tee $O/ab_errlog_swar.log
timeout 300 python scripts/parity_sweep.py 30000 2>&1 | tail -4 | tee $O/parity_sweep.log
