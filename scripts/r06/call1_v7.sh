#!/bin/bash
# round 6, GPU call 1: record-kernel tile v7 (chunk lanes load their own sub-run) against v6 (sub-run elements): same-box A/B of the default
# line incl. configs[2], then the parity suite on v7.
cd "$(dirname "$0")/../.."
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r06a; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
for name in ${VARIANTS:-v6 v7 v6 v7}; do
  f=nanosim_amd/_variants/$name.so
  NANOSIM_AMD_LIB=$PWD/$f timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-e2e --extras-steps 3 2>$O/err_$name.log | tail -1 > $O/bench_$name.json
  python - $name $O/bench_$name.json <<'P' | tee -a $O/ab.log
import json,sys
name,p=sys.argv[1:3]
try:
    d=json.load(open(p))
except Exception as ex:
    print(name,"FAILED",ex); sys.exit(0)
r=lambda x:round(x,2)
s=d.get("serial",{}); e=d.get("errlog_on",{}); c=d.get("configs2",{})
print(name,"step",r(d["ms_per_step"]),"ms",r(d["value"]/1e6),"M/s | aligned",{k:r(v) for k,v in d["kernel_ms"].items() if v>0.01},"frac",r(d["roofline"]["frac"]),
      "| serial",r(s.get("ms_per_step",0)),"al",{k:r(v) for k,v in (s.get("aligned_kernel_ms") or {}).items() if v>0.01},
      "| errlog_on",r(e.get("ms_per_step",0)),
      "| configs2",r(c.get("ms_per_step",0)),{k:r(v) for k,v in (c.get("aligned_batch",{}).get("kernel_ms") or {}).items() if v>0.01})
P
done
( timeout 60 python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -2; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 ) | tee $O/pytest_gpu.log
