#!/bin/bash
# round 6, GPU call 5: k_chain occupancy variants (six waves per SIMD: 80 VGPRs with scratch; 512-thread workgroups sharing one table image;
# NS_NO_EV_STAGE=1: no LDS event staging, i.e. 24 instead of 32 KB per 256-thread workgroup), same-box A/B on the default workload
cd "$(dirname "$0")/../.."
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r06e; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
run() {
  name=$1; tag=$2; shift 2
  f=nanosim_amd/_variants/$name.so
  env "$@" NANOSIM_AMD_LIB=$PWD/$f timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-e2e --no-configs2 --extras-steps 3 2>$O/err_$tag.log | tail -1 > $O/bench_$tag.json
  python - $tag $O/bench_$tag.json <<'P' | tee -a $O/ab.log
import json,sys
name,p=sys.argv[1:3]
try:
    d=json.load(open(p))
except Exception as ex:
    print(name,"FAILED",ex); sys.exit(0)
r=lambda x:round(x,2)
s=d.get("serial",{})
print(name,"step",r(d["ms_per_step"]),"ms",r(d["value"]/1e6),"M/s | aligned",{k:r(v) for k,v in d["kernel_ms"].items() if v>0.01},
      "| serial",r(s.get("ms_per_step",0)),"al",{k:r(v) for k,v in (s.get("aligned_kernel_ms") or {}).items() if v>0.01},"un",{k:r(v) for k,v in (s.get("unaligned_kernel_ms") or {}).items() if v>0.01})
P
}
run base base X=1
run c6 c6 X=1
run c6 c6_nostage NS_NO_EV_STAGE=1
run base base_nostage NS_NO_EV_STAGE=1
run c6b512 c6b512 X=1
run c5b512 c5b512 X=1
run base base2 X=1
( timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "every_k or circular or overflow or capacity" 2>&1 | tail -4 ) | tee $O/pytest.log
