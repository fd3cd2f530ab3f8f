#!/bin/bash
# round 6, GPU call 50: what k_names costs a step: a build whose k_names returns at once under NS_DEBUG_SKIP bit 23 (profiling only), alternating
cd "$(dirname "$0")/../.."
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r06be; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0 NANOSIM_AMD_LIB=$PWD/nanosim_amd/_variants/nonames.so
for rep in 1 2 3 4; do for sk in 0 8388608; do
  NS_DEBUG_SKIP=$sk timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-e2e --no-configs2 --no-extras 2>$O/err.log | tail -1 > $O/b.json
  python - "skip=$sk" $O/b.json <<'P' | tee -a $O/ab.log
import json,sys
name,p=sys.argv[1:3]
d=json.load(open(p)); r=lambda x:round(x,3); a=d["aligned_batch"]; k=a["kernel_ms"]
print("%-14s"%name,"step",r(d["ms_per_step"]),"ms | aligned call",r(a["device_ms"]),"chain",r(k["k_chain"]),"record stage",r(k["k_materialise"]),"kernel",r(d["roofline"]["kernel_ms"]))
P
done; done
