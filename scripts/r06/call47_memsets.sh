#!/bin/bash
# round 6, GPU call 47: the 4-byte memsets in front of the chain and the record kernel folded into k_stats_fold / dropped for pass 0 (new) against the
# commit before (base; scripts: variants built from HEAD and from the working tree): parity, then alternating bench lines
cd "$(dirname "$0")/../.."
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r06ba; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0

for rep in 1 2 3 4; do for v in base new; do
  NANOSIM_AMD_LIB=$PWD/nanosim_amd/_variants/$v.so timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-e2e --no-configs2 --no-extras 2>$O/err.log | tail -1 > $O/b.json
  python - "$v" $O/b.json <<'P' | tee -a $O/ab.log
import json,sys
name,p=sys.argv[1:3]
try:
    d=json.load(open(p)); r=lambda x:round(x,3); u=d["unaligned_batch"]; a=d["aligned_batch"]; k=a["kernel_ms"]
    gap=a["device_ms"]-k["plan(k_nseg+k_lengths+scan+sort)"]-k["k_chain"]-k["k_materialise"]
    print("%-6s"%name,"step",r(d["ms_per_step"]),"ms",r(d["value"]/1e6),"M/s frac",r(d["roofline"]["frac"]),"| aligned call",r(a["device_ms"]),"chain",r(k["k_chain"]),"record stage",r(k["k_materialise"]),"kernel",r(d["roofline"]["kernel_ms"]),"gaps",r(gap),
          "| unaligned call",r(u["device_ms"]))
except Exception as ex:
    print(name,"FAILED",ex)
P
done; done
