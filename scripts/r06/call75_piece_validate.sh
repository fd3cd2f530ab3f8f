#!/bin/bash
# round 6, GPU call 75: the thread-per-piece build: the parity suites of all modes (k_lengths changed for every mode), then configs[3] / [1] lines
cd "$(dirname "$0")/../.."
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r06pv; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
( timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 ) | tee $O/pytest.log
for cfg in "grch38 --chimeric" "chr1 --chimeric" "ecoli "; do
  set -- $cfg
  timeout 400 python bench.py --genome $1 $2 --steps 6 --warmup 2 --no-cpu-baseline --no-e2e --no-configs2 --extras-steps 3 2>$O/err.log | tail -1 > $O/b.json
  python - "$1 $2" $O/b.json <<'P' | tee -a $O/ab.log
import json,sys
name,p=sys.argv[1:3]
d=json.load(open(p)); r=lambda x:round(x,3); a=d["aligned_batch"]; k=a["kernel_ms"]; s=d.get("serial",{})
print("%-24s"%name,"step",r(d["ms_per_step"]),"ms",r(d["value"]/1e6),"M/s | aligned call",r(a["device_ms"]),"plan",r(k["plan(k_nseg+k_lengths+scan+sort)"]),"chain",r(k["k_chain"]),"record",r(k["k_materialise"]),"| chain alone",r((s.get("aligned_kernel_ms") or {}).get("k_chain",0)),"| errlog_on step",r(d.get("errlog_on",{}).get("ms_per_step",0)))
P
done
