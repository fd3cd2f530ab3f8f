#!/bin/bash
# round 6, GPU call 69: chimeric batches — issue priority of the thread-per-read chain by planned work instead of by position (NS_PRIO_BY_POSITION=1: as before)
cd "$(dirname "$0")/../.."
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r06bz; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
( timeout 1800 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3 ) | tee $O/pytest.log
run() {
  timeout 400 python bench.py --genome $1 $2 --steps 6 --warmup 2 --no-cpu-baseline --no-e2e --no-configs2 --extras-steps 3 2>$O/err.log | tail -1 > $O/b.json
  python - "$1 $2 shift=${NS_COOP_MULTI_SHIFT:-def} bypos=${NS_PRIO_BY_POSITION:-0}" $O/b.json <<'P' | tee -a $O/ab.log
import json,sys
name,p=sys.argv[1:3]
d=json.load(open(p)); r=lambda x:round(x,3); a=d["aligned_batch"]; k=a["kernel_ms"]; s=d.get("serial",{})
print("%-40s"%name,"step",r(d["ms_per_step"]),"ms",r(d["value"]/1e6),"M/s | aligned call",r(a["device_ms"]),"plan",r(k["plan(k_nseg+k_lengths+scan+sort)"]),"chain",r(k["k_chain"]),"record",r(k["k_materialise"]),"| chain alone",r((s.get("aligned_kernel_ms") or {}).get("k_chain",0)))
P
}
run chr1 ""
NS_PRIO_BY_POSITION=1 run chr1 --chimeric
run chr1 --chimeric
NS_PRIO_BY_POSITION=1 run chr1 --chimeric
run chr1 --chimeric
for sh in 7 10; do NS_COOP_MULTI_SHIFT=$sh run chr1 --chimeric; done
run grch38 --chimeric
export NANOSIM_AMD_LIB=$PWD/build_ab/libns_clock.so
timeout 400 python bench.py --genome chr1 --chimeric --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-configs2 --no-extras 2>$O/err.log >/dev/null
grep "chain clock" $O/err.log | tail -2 | tee -a $O/ab.log
