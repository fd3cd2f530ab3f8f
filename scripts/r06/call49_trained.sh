#!/bin/bash
# round 6, GPU call 49: the trained-model table shape inside the step as it is now (the unaligned call's chain next to it): workgroup size and prefix length again
cd "$(dirname "$0")/../.."
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r06bd; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
for rep in 1 2; do for cfg in "- -" "256 -" "- 10" "256 10" "256 8" "- 14"; do
  set -- $cfg
  unset NS_CHAIN_BLOCK NS_TAIL_BITS
  [ "$1" != "-" ] && export NS_CHAIN_BLOCK=$1
  [ "$2" != "-" ] && export NS_TAIL_BITS=$2
  timeout 300 python bench.py --trained-shape --steps 8 --warmup 3 --no-cpu-baseline --no-e2e --no-configs2 --extras-steps 3 2>$O/err.log | tail -1 > $O/b.json
  python - "block=$1 tail_bits=$2" $O/b.json <<'P' | tee -a $O/ab.log
import json,sys
name,p=sys.argv[1:3]
try:
    d=json.load(open(p)); r=lambda x:round(x,2); u=d["unaligned_batch"]; a=d["aligned_batch"]; s=d.get("serial",{})
    print("%-26s"%name,"step",r(d["ms_per_step"]),"ms",r(d["value"]/1e6),"M/s | aligned call",r(a["device_ms"]),"chain",r(a["kernel_ms"]["k_chain"]),"record",r(a["kernel_ms"]["k_materialise"]),"| chain alone",r((s.get("aligned_kernel_ms") or {}).get("k_chain",0)))
except Exception as ex:
    print(name,"FAILED",ex)
P
done; done
