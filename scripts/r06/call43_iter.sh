#!/bin/bash
# (variants first, here: scripts/ab_build.sh k4:"-DNS_UCOOP_ITER=4" k3:"-DNS_UCOOP_ITER=3" k2:"-DNS_UCOOP_ITER=2" — second run: k3 k3w5:"-DNS_UCOOP_ITER=3 -DNS_UCOOP_MINW=5"
#  k4w5:"-DNS_UCOOP_ITER=4 -DNS_UCOOP_MINW=5" k2w6:"-DNS_UCOOP_ITER=2 -DNS_UCOOP_MINW=6")
# round 6, GPU call 43: loop iterations per lane of the wave-per-read unaligned chain (4 / 3 / 2: 116 / 98 / 82 VGPRs) — alone and inside the step
cd "$(dirname "$0")/../.."
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r06aw; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
for v in k3 k3w5 k4w5 k2w6; do echo "== $v alone"; NANOSIM_AMD_LIB=$PWD/nanosim_amd/_variants/$v.so K=3 timeout 200 python scripts/r06/unaligned_probe.py 2>/dev/null | tail -2; done | tee $O/alone.log
for rep in 1 2 3; do for v in k3 k3w5 k4w5 k2w6; do
  NANOSIM_AMD_LIB=$PWD/nanosim_amd/_variants/$v.so timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-e2e --no-configs2 --no-extras 2>$O/err.log | tail -1 > $O/b.json
  python - "$v" $O/b.json <<'P' | tee -a $O/ab.log
import json,sys
name,p=sys.argv[1:3]
try:
    d=json.load(open(p)); r=lambda x:round(x,2); u=d["unaligned_batch"]; a=d["aligned_batch"]
    print("%-6s"%name,"step",r(d["ms_per_step"]),"ms",r(d["value"]/1e6),"M/s frac",round(d["roofline"]["frac"],3),"| aligned call",r(a["device_ms"]),"chain",r(a["kernel_ms"]["k_chain"]),"record",r(a["kernel_ms"]["k_materialise"]),
          "| unaligned call",r(u["device_ms"]),"chain",r(u["kernel_ms"]["k_chain"]),"dense",r(u["kernel_ms"]["k_materialise"]))
except Exception as ex:
    print(name,"FAILED",ex)
P
done; done
