#!/bin/bash
# (variants: scripts/ab_build.sh n256:"-DNS_NAMES_BLOCK=256" n64:"-DNS_NAMES_BLOCK=64" n128:"-DNS_NAMES_BLOCK=128")
# round 6, GPU call 52: workgroup size of k_names (its 256-thread workgroups with 32 KB of LDS wait for room until the record kernel drains:
# it ends 30-50 us after it)
cd "$(dirname "$0")/../.."
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r06bh; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
for rep in 1 2 3 4; do for v in n256 n64 n128; do
  NANOSIM_AMD_LIB=$PWD/nanosim_amd/_variants/$v.so timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-e2e --no-configs2 --no-extras 2>$O/err.log | tail -1 > $O/b.json
  python - "$v" $O/b.json <<'P' | tee -a $O/ab.log
import json,sys
name,p=sys.argv[1:3]
d=json.load(open(p)); r=lambda x:round(x,3); a=d["aligned_batch"]; k=a["kernel_ms"]
print("%-6s"%name,"step",r(d["ms_per_step"]),"ms",r(d["value"]/1e6),"M/s frac",r(d["roofline"]["frac"]),"| aligned call",r(a["device_ms"]),"chain",r(k["k_chain"]),"record stage",r(k["k_materialise"]),"kernel",r(d["roofline"]["kernel_ms"]))
P
done; done
