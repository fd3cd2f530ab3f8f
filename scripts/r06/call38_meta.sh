#!/bin/bash
# round 6, GPU call 38: metagenome passes with the error lists launched before the host walks the quotas (k_meta_tail): parity, the configs[4] line
cd "$(dirname "$0")/../.."
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r06ao; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
( timeout 1200 python -m pytest tests/test_gpu_metagenome.py tests/test_gpu_cli.py -m gpu -x -q 2>&1 | tail -5 ) | tee $O/pytest.log
for rep in 1 2; do
timeout 300 python bench.py --metagenome --steps 6 --warmup 2 --no-cpu-baseline --no-e2e 2>$O/bench_meta.err | tail -1 > $O/bench_zymo10_metagenome.json
python -c "
import json,sys
d=json.load(open('$O/bench_zymo10_metagenome.json')); r=lambda x:round(x,3)
print('zymo10', r(d['ms_per_step']), 'ms', r(d['value']/1e6), 'M reads/s', {k:r(v) for k,v in d['kernel_ms'].items() if v>0.01}, 'aligned call', r(d['aligned_batch']['device_ms']), 'unaligned', r(d['unaligned_batch']['device_ms']), 'serial', r(d.get('serial',{}).get('ms_per_step',0)), 'errlog_on', r(d.get('errlog_on',{}).get('ms_per_step',0)))
" | tee -a $O/summary.log
done
timeout 300 python bench.py --metagenome --chimeric --steps 6 --warmup 2 --no-cpu-baseline --no-e2e --no-extras 2>>$O/bench_meta.err | tail -1 > $O/b2.json
python -c "
import json,sys
d=json.load(open('$O/b2.json')); r=lambda x:round(x,3)
print('zymo10 --chimeric', r(d['ms_per_step']), 'ms', r(d['value']/1e6), 'M reads/s', {k:r(v) for k,v in d['kernel_ms'].items() if v>0.01})
" | tee -a $O/summary.log
NS_META_TRACE=1 timeout 300 python bench.py --metagenome --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-extras --aligned-only 2>&1 >/dev/null | grep "\[meta\]" | tail -14 | tee $O/trace.log
