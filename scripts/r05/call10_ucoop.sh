#!/bin/bash
# round 5, GPU call 10: the wave-per-read UNALIGNED chain with its run-length tables in LDS (k_chain<true, true>, NS_UCOOP_LDS=1 — default)
# against the tables in global memory (NS_UCOOP_LDS=0: rounds 2-5), and which share of the unaligned reads of the step companion should take
# it (NS_UCOOP_SHIFT: longest n >> shift; 0 = all).  Parity first: the unaligned reads of the GPU parity tests go through the new kernel.
cd "$(dirname "$0")/../.."
O=gpurun_out/r05j; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -3 | tee $O/pytest_parity.log
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=lambda x:round(x,2)
s=d.get('serial',{})
print(r(d['ms_per_step']), 'ms/step', r(d['value']/1e6), 'M reads/s; aligned', r(d['aligned_batch']['device_ms']), {k:r(v) for k,v in d['kernel_ms'].items() if v>0.01}, '; unaligned', r(d['unaligned_batch']['device_ms']), {k:r(v) for k,v in d['unaligned_batch']['kernel_ms'].items() if v>0.01}, '; serial', r(s.get('ms_per_step',0)), 'unaligned alone', r(s.get('unaligned_device_ms',0)), {k:r(v) for k,v in s.get('unaligned_kernel_ms',{}).items() if v>0.01})"; }
for g in "NS_UCOOP_LDS=0" "NS_UCOOP_LDS=1"; do
  echo -n "$g: "; env $g timeout 240 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-e2e --no-configs2 2>/dev/null | tail -1 | line; done 2>&1 | tee $O/ab_ucoop_lds.log
for g in "NS_UCOOP_LDS=1 NS_UCOOP_SHIFT=2" "NS_UCOOP_LDS=1 NS_UCOOP_SHIFT=1" "NS_UCOOP_LDS=1 NS_UCOOP_SHIFT=0" "NS_UCOOP_LDS=0" "NS_UCOOP_LDS=1"; do
  echo -n "$g: "; env $g timeout 200 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-e2e --no-configs2 --no-extras 2>/dev/null | tail -1 | line; done 2>&1 | tee -a $O/ab_ucoop_lds.log
