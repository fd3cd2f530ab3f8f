#!/bin/bash
# round 6, GPU call 9: k_errlog of the current build against the round-5 library (same box): the r06 profile run showed 6.5-6.6 ms where r05 recorded 5.9-6.1
cd "$(dirname "$0")/../.."
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r06i; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
for name in r05 cur r05 cur; do
  f=nanosim_amd/_variants/$name.so
  NANOSIM_AMD_LIB=$PWD/$f timeout 300 python bench.py --errlog --aligned-only --steps 6 --warmup 2 --no-cpu-baseline --no-e2e --no-extras 2>$O/err_$name.log | tail -1 > $O/bench_$name.json
  NANOSIM_AMD_LIB=$PWD/$f timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-e2e --no-configs2 --extras-steps 4 2>$O/err2_$name.log | tail -1 > $O/bench2_$name.json
  python - $name $O/bench_$name.json $O/bench2_$name.json <<'P' | tee -a $O/ab.log
import json,sys
name,p,p2=sys.argv[1:4]
r=lambda x:round(x,2)
try:
    d=json.load(open(p)); print(name,"aligned-only --errlog step",r(d["ms_per_step"]),"ms |",{k:r(v) for k,v in d["kernel_ms"].items() if v>0.01})
    d=json.load(open(p2)); e=d["errlog_on"]; print(name,"default step",r(d["ms_per_step"]),{k:r(v) for k,v in d["kernel_ms"].items() if v>0.01},"errlog_on",r(e["ms_per_step"]),"k_errlog",r(e["k_errlog_ms"]))
except Exception as ex:
    print(name,"FAILED",ex)
P
done
