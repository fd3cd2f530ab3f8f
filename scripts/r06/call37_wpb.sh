#!/bin/bash
# (needs a build with NS_UCOOP_WPB: commit 89c8eb7 describes it; the macro is not in the tree any more)
# round 6, GPU call 37: reads per workgroup of the wave-per-read unaligned chain (1 / 4 / 8 wavefronts sharing the LDS tables): parity, the call
# alone, the kernel without its lists (floor), the step
cd "$(dirname "$0")/../.."
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r06an; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "unaligned or background or gpu_equals_oracle" 2>&1 | tail -3 ) | tee $O/pytest.log
for v in wpb1 wpb4 wpb8; do echo "== $v alone"; NANOSIM_AMD_LIB=$PWD/nanosim_amd/_variants/$v.so K=3 timeout 200 python scripts/r06/unaligned_probe.py 2>/dev/null | tail -2; done | tee $O/alone.log
echo "== wpb4, -DNS_ABLATE, without the error lists" | tee -a $O/alone.log
NANOSIM_AMD_LIB=$PWD/nanosim_amd/_variants/abl4.so NS_DEBUG_SKIP=$((1<<20)) K=2 timeout 200 python scripts/r06/unaligned_probe.py 2>/dev/null | tail -1 | tee -a $O/alone.log
for rep in 1 2; do for v in wpb1 wpb4 wpb8; do for sh in 3 0; do
  NANOSIM_AMD_LIB=$PWD/nanosim_amd/_variants/$v.so NS_UCOOP_SHIFT=$sh timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-e2e --no-configs2 --extras-steps 4 2>$O/err.log | tail -1 > $O/b.json
  python - "$v ucoop_shift=$sh" $O/b.json <<'P' | tee -a $O/ab.log
import json,sys
name,p=sys.argv[1:3]
try:
    d=json.load(open(p)); r=lambda x:round(x,2); u=d["unaligned_batch"]; a=d["aligned_batch"]; s=d.get("serial",{})
    print("%-24s"%name,"step",r(d["ms_per_step"]),"ms",r(d["value"]/1e6),"M/s | aligned call",r(a["device_ms"]),"chain",r(a["kernel_ms"]["k_chain"]),"record",r(a["kernel_ms"]["k_materialise"]),
          "| unaligned call",r(u["device_ms"]),"chain",r(u["kernel_ms"]["k_chain"]),"dense",r(u["kernel_ms"]["k_materialise"]),"| serial",r(s.get("ms_per_step",0)))
except Exception as ex:
    print(name,"FAILED",ex)
P
done; done; done
