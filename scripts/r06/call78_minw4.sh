#!/bin/bash
# round 6, GPU call 78: the thread-per-read chain compiled for four wavefronts per SIMD (109 VGPRs, no scratch) against five (96 VGPRs, 17 spilled): the launch is
# bound by its slowest wavefront (call 68), which fewer neighbours and no scratch traffic might speed up (build_ab/libns_minw4.so: -DNS_CHAIN_MINW=4)
cd "$(dirname "$0")/../.."
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r06mw; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
run() {
  timeout 400 python bench.py --genome $1 $2 --steps 8 --warmup 2 --no-cpu-baseline --no-e2e --no-configs2 --extras-steps 3 2>$O/err.log | tail -1 > $O/b.json
  python - "$1 $2 $3" $O/b.json <<'P' | tee -a $O/ab.log
import json,sys
name,p=sys.argv[1:3]
d=json.load(open(p)); r=lambda x:round(x,3); a=d["aligned_batch"]; k=a["kernel_ms"]; s=d.get("serial",{})
print("%-40s"%name,"step",r(d["ms_per_step"]),"ms",r(d["value"]/1e6),"M/s | aligned call",r(a["device_ms"]),"chain",r(k["k_chain"]),"record",r(k["k_materialise"]),"| chain alone",r((s.get("aligned_kernel_ms") or {}).get("k_chain",0)),"| trained shape step",r(d.get("trained_shape",{}).get("ms_per_step",0)))
P
}
for rep in 1 2 3; do
  run ecoli "" "five waves"
  NANOSIM_AMD_LIB=$PWD/build_ab/libns_minw4.so run ecoli "" "four waves"
done
