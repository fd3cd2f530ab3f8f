#!/bin/bash
# round 6, GPU call 21: the slow look-ups of the chain by bisection between the guide's bounds (ecdf_lookup_pre / ecdf_lookup_gv) against the linear
# walks of the commit before, alternating, default model and trained shape; chain alone = the serial object
cd "$(dirname "$0")/../.."
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r06u; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
for rep in 1 2 3; do for name in walk bisect; do for X in "" "--trained-shape"; do
  f=nanosim_amd/_variants/$name.so
  NANOSIM_AMD_LIB=$PWD/$f timeout 300 python bench.py $X --steps 6 --warmup 2 --no-cpu-baseline --no-e2e --no-configs2 --extras-steps 4 2>$O/err.log | tail -1 > $O/b.json
  python - "$name$X" $O/b.json <<'P' | tee -a $O/ab.log
import json,sys
name,p=sys.argv[1:3]
try:
    d=json.load(open(p)); r=lambda x:round(x,2); s=d.get("serial",{})
    print("%-24s"%name,"step",r(d["ms_per_step"]),"ms",r(d["value"]/1e6),"M/s | in-step chain",r(d["kernel_ms"]["k_chain"]),"record",r(d["kernel_ms"]["k_materialise"]),"| chain alone",r((s.get("aligned_kernel_ms") or {}).get("k_chain",0)))
except Exception as ex:
    print(name,"FAILED",ex)
P
done; done; done
