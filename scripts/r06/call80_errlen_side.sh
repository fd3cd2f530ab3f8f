#!/bin/bash
# round 6, GPU call 80: the sizes of the error-profile rows (k_errlen + scan) on the second stream next to the record kernel instead of in front of it
# (NS_ERRLEN_INLINE=1: as before): parity of everything that writes an error profile, then the errlog_on step, alternating
cd "$(dirname "$0")/../.."
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r06el; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
( timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sinks.py tests/test_gpu_cli.py tests/test_gpu_fullsize.py tests/test_gpu_fullsize_refs.py -m gpu -x -q 2>&1 | tail -3 ) | tee $O/pytest.log
run() {
  timeout 400 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-e2e --no-configs2 --extras-steps 5 2>$O/err.log | tail -1 > $O/b.json
  python - "inline=${NS_ERRLEN_INLINE:-0}" $O/b.json <<'P' | tee -a $O/ab.log
import json,sys
name,p=sys.argv[1:3]
d=json.load(open(p)); r=lambda x:round(x,3); e=d["errlog_on"]
print("%-10s"%name,"step",r(d["ms_per_step"]),"ms | errlog_on step",r(e["ms_per_step"]),"ms",r(e["value"]/1e6),"M reads/s aligned call",r(e["aligned_device_ms"]),"k_errlog",r(e["k_errlog_ms"]))
P
}
for rep in 1 2 3; do NS_ERRLEN_INLINE=1 run; run; done
