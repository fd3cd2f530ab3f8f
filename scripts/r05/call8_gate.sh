#!/bin/bash
# round 5, GPU call 8: does holding the companion's chain launch until the aligned call has launched its own (NS_STEP_GATE=1) shorten the
# step?  (the aligned call's planning kernels take 0.72 ms next to the unaligned chain against 0.29 ms alone); then the configs[3] /
# configs[4] lines with ns_generate_step
cd "$(dirname "$0")/../.."
O=gpurun_out/r05i; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
for g in "" "NS_STEP_GATE=1" "" "NS_STEP_GATE=1"; do echo -n "${g:-no gate} "; env $g timeout 150 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-e2e --no-configs2 --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=lambda x:round(x,2)
print(r(d['ms_per_step']), 'ms/step', r(d['value']/1e6), 'M reads/s; aligned', r(d['aligned_batch']['device_ms']), {k:r(v) for k,v in d['kernel_ms'].items() if v>0.01}, '; unaligned', r(d['unaligned_batch']['device_ms']), {k:r(v) for k,v in d['unaligned_batch']['kernel_ms'].items() if v>0.01})"; done 2>&1 | tee $O/ab_step_gate.log
timeout 400 python bench.py --genome grch38 --chimeric --steps 5 --warmup 2 --no-cpu-baseline --no-e2e 2>$O/bench_grch38.err | tail -1 > $O/bench_grch38_chimeric.json
timeout 300 python bench.py --metagenome --steps 5 --warmup 2 --no-cpu-baseline --no-e2e 2>$O/bench_meta.err | tail -1 > $O/bench_zymo10_metagenome.json
for f in bench_grch38_chimeric bench_zymo10_metagenome; do python -c "
import json,sys
d=json.load(open('$O/$f.json')); r=lambda x:round(x,3)
print('$f', r(d['ms_per_step']), 'ms', r(d['value']/1e6), 'M reads/s', {k:r(v) for k,v in d['kernel_ms'].items() if v>0.01}, 'serial', r(d.get('serial',{}).get('ms_per_step',0)), 'errlog_on', r(d.get('errlog_on',{}).get('ms_per_step',0)), r(d.get('errlog_on',{}).get('k_errlog_ms',0)), 'frac', r(d['roofline']['frac']))
" | tee -a $O/bench_other_summary.log; done
