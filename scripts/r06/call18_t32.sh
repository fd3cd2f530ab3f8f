#!/bin/bash
# round 6, GPU call 18: the prefix columns as 32-bit thresholds (image 15 KB for the bench model, 30 KB for the trained shape): parity, then
# default / trained-shape lines alternating, then workgroup sizes for the trained shape
cd "$(dirname "$0")/../.."
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r06t; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_metagenome.py tests/test_gpu_transcriptome.py -m gpu -x -q 2>&1 | tail -4 ) | tee $O/pytest_parity.log
run() { tag=$1; shift
  env "$@" timeout 300 python bench.py $X --steps 6 --warmup 2 --no-cpu-baseline --no-e2e --no-configs2 --extras-steps 3 2>$O/err_$tag.log | tail -1 > $O/bench_$tag.json
  python - $tag $O/bench_$tag.json <<'P' | tee -a $O/ab.log
import json,sys
name,p=sys.argv[1:3]
try:
    d=json.load(open(p))
except Exception as ex:
    print(name,"FAILED",ex); sys.exit(0)
r=lambda x:round(x,2)
s=d.get("serial",{})
print(name,"step",r(d["ms_per_step"]),"ms",r(d["value"]/1e6),"M/s | aligned",{k:r(v) for k,v in d["kernel_ms"].items() if v>0.01},"frac",r(d["roofline"]["frac"]),
      "| serial",r(s.get("ms_per_step",0)),"al",{k:r(v) for k,v in (s.get("aligned_kernel_ms") or {}).items() if v>0.01},"un",{k:r(v) for k,v in (s.get("unaligned_kernel_ms") or {}).items() if v>0.01})
P
}
X=""; run default Y=1
X="--trained-shape"; run trained Y=1
X=""; run default Y=1
X="--trained-shape"; run trained Y=1; run trained_256 NS_CHAIN_BLOCK=256; run trained_tb14 NS_TAIL_BITS=14; run trained_tb10 NS_TAIL_BITS=10; run trained_global NS_TAIL_BITS=31
