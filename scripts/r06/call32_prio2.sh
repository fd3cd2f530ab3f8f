#!/bin/bash
# round 6, GPU call 32: the companion's streams at low / default / high priority, and at the default priority with eight hardware queues
cd "$(dirname "$0")/../.."
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r06ai; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
export NS_UWIDE_SHIFT=31
for rep in 1 2; do for cfg in "0 4" "0 8" "-1 4" "1 4" "-1 8"; do
  set -- $cfg
  NS_STEP_PRIO=$1 GPU_MAX_HW_QUEUES=$2 timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-e2e --no-configs2 --no-extras 2>$O/err.log | tail -1 > $O/b.json
  python - "prio=$1 hw_queues=$2" $O/b.json <<'P' | tee -a $O/ab.log
import json,sys
name,p=sys.argv[1:3]
try:
    d=json.load(open(p)); r=lambda x:round(x,2); u=d["unaligned_batch"]; a=d["aligned_batch"]
    print("%-24s"%name,"step",r(d["ms_per_step"]),"ms",r(d["value"]/1e6),"M/s | aligned call",r(a["device_ms"]),"chain",r(a["kernel_ms"]["k_chain"]),"record",r(a["kernel_ms"]["k_materialise"]),
          "| unaligned call",r(u["device_ms"]),"chain",r(u["kernel_ms"]["k_chain"]),"dense",r(u["kernel_ms"]["k_materialise"]))
except Exception as ex:
    print(name,"FAILED",ex)
P
done; done
