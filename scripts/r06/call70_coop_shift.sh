#!/bin/bash
# round 6, GPU call 70: the share of the aligned reads on the wave-per-read list (NS_COOP_SHIFT) on the last build — the thread-per-read launch is bound by its
# slowest wavefront (call 68), the wave-per-read kernel got faster since the sweep of call 23
cd "$(dirname "$0")/../.."
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r06ca; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
run() {
  timeout 400 python bench.py --genome $1 $2 --steps 8 --warmup 2 --no-cpu-baseline --no-e2e --no-configs2 --extras-steps 3 2>$O/err.log | tail -1 > $O/b.json
  python - "$1 $2 NS_COOP_SHIFT=${NS_COOP_SHIFT:-def}" $O/b.json <<'P' | tee -a $O/ab.log
import json,sys
name,p=sys.argv[1:3]
d=json.load(open(p)); r=lambda x:round(x,3); a=d["aligned_batch"]; k=a["kernel_ms"]; s=d.get("serial",{})
print("%-40s"%name,"step",r(d["ms_per_step"]),"ms",r(d["value"]/1e6),"M/s | aligned call",r(a["device_ms"]),"plan",r(k["plan(k_nseg+k_lengths+scan+sort)"]),"chain",r(k["k_chain"]),"record",r(k["k_materialise"]),"| chain alone",r((s.get("aligned_kernel_ms") or {}).get("k_chain",0)))
P
}
for rep in 1 2; do for sh in 10 9 8 11; do NS_COOP_SHIFT=$sh run ecoli ""; done; done
