#!/bin/bash
# round 6, GPU call 45: stream priority of the step companion again (NS_STEP_PRIO -1 / 0 / 1), now that its call is short
cd "$(dirname "$0")/../.."
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r06ay; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
for rep in 1 2 3; do for pr in 0 -1 1; do
  NS_STEP_PRIO=$pr timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-e2e --no-configs2 --no-extras 2>$O/err.log | tail -1 > $O/b.json
  python - "prio=$pr" $O/b.json <<'P' | tee -a $O/ab.log
import json,sys
name,p=sys.argv[1:3]
try:
    d=json.load(open(p)); r=lambda x:round(x,2); u=d["unaligned_batch"]; a=d["aligned_batch"]
    print("%-8s"%name,"step",r(d["ms_per_step"]),"ms",r(d["value"]/1e6),"M/s frac",round(d["roofline"]["frac"],3),"| aligned call",r(a["device_ms"]),"chain",r(a["kernel_ms"]["k_chain"]),"record",r(a["kernel_ms"]["k_materialise"]),
          "| unaligned call",r(u["device_ms"]),"chain",r(u["kernel_ms"]["k_chain"]),"dense",r(u["kernel_ms"]["k_materialise"]))
except Exception as ex:
    print(name,"FAILED",ex)
P
done; done
