#!/bin/bash
# round 6, GPU call 76: a chimeric step now ends with the unaligned companion (0.3 ms after the aligned call): its stream priority (NS_STEP_PRIO) and its
# wave-per-read share (NS_UCOOP_SHIFT) on the thread-per-piece build
cd "$(dirname "$0")/../.."
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r06cc; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
run() {
  timeout 400 python bench.py --genome $1 $2 --steps 8 --warmup 2 --no-cpu-baseline --no-e2e --no-configs2 --no-extras 2>$O/err.log | tail -1 > $O/b.json
  python - "$1 $2 NS_STEP_PRIO=${NS_STEP_PRIO:-0} NS_STEP_GATE=${NS_STEP_GATE:-1}" $O/b.json <<'P' | tee -a $O/ab.log
import json,sys
name,p=sys.argv[1:3]
d=json.load(open(p)); r=lambda x:round(x,3); a=d["aligned_batch"]; k=a["kernel_ms"]; u=d["unaligned_batch"]
print("%-50s"%name,"step",r(d["ms_per_step"]),"ms",r(d["value"]/1e6),"M/s | aligned call",r(a["device_ms"]),"chain",r(k["k_chain"]),"record",r(k["k_materialise"]),"| unaligned call",r(u["device_ms"]),{x:r(v) for x,v in u["kernel_ms"].items() if v>0.05})
P
}
for rep in 1 2; do
run grch38 --chimeric
NS_STEP_PRIO=1 run grch38 --chimeric
NS_STEP_GATE=0 run grch38 --chimeric
done
