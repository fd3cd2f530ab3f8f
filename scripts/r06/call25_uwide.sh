#!/bin/bash
# (the workgroup-per-read kernel and NS_UWIDE_SHIFT were removed again: commit a56cb35 has the log)
# round 6, GPU call 25: the longest unaligned reads of a batch on a workgroup of 16 wavefronts each (NS_UWIDE_SHIFT; 31 = off = the build before):
# parity, then the step with the share at several values
cd "$(dirname "$0")/../.."
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r06ab; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "workgroup_per_read or background_context or unaligned" 2>&1 | tail -4 ) | tee $O/pytest_uwide.log
for rep in 1 2; do for sh in 31 6 4 8; do
  NS_UWIDE_SHIFT=$sh timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-e2e --no-configs2 --extras-steps 4 2>$O/err.log | tail -1 > $O/b.json
  python - "uwide_shift=$sh" $O/b.json <<'P' | tee -a $O/ab.log
import json,sys
name,p=sys.argv[1:3]
try:
    d=json.load(open(p)); r=lambda x:round(x,2); s=d.get("serial",{}); u=d["unaligned_batch"]; a=d["aligned_batch"]
    print("%-16s"%name,"step",r(d["ms_per_step"]),"ms",r(d["value"]/1e6),"M/s | aligned call",r(a["device_ms"]),"chain",r(a["kernel_ms"]["k_chain"]),"record",r(a["kernel_ms"]["k_materialise"]),
          "| unaligned call",r(u["device_ms"]),"chain",r(u["kernel_ms"]["k_chain"]),"| serial",r(s.get("ms_per_step",0)),"unaligned chain alone",r((s.get("unaligned_kernel_ms") or {}).get("k_chain",0)))
except Exception as ex:
    print(name,"FAILED",ex)
P
done; done
tail -3 $O/err.log
