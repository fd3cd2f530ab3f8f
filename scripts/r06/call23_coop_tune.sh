#!/bin/bash
# round 6, GPU call 23: after the wave-per-read chain got faster — kernel durations (rocprofv3) of the bulk and the wave-per-read chain, then the share of
# reads on the wave-per-read list (NS_COOP_SHIFT: longest n >> shift), default model and trained shape
cd "$(dirname "$0")/../.."
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06z; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd /tmp; export TMPDIR=/tmp
for X in "" "--trained-shape"; do
  rm -rf /tmp/ks; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -o s -- python $R/bench.py $X --aligned-only --steps 4 --warmup 1 --no-cpu-baseline --no-e2e --no-configs2 --no-extras > /dev/null 2>&1
  python3 - "$X" <<'P' | tee -a $O/kernels.log
import csv,glob,sys
for f in glob.glob('/tmp/ks/**/*kernel_stats.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        if 'k_chain' in row['Name'] or 'k_materialise<' in row['Name']: print(sys.argv[1] or "default", row['Name'][:40], row['Calls'], round(float(row['AverageNs'])/1e6,3), 'max', round(float(row['MaxNs'])/1e6,3))
P
done
cd $R
for sh in 10 9 8 7; do for X in "" "--trained-shape"; do
  NS_COOP_SHIFT=$sh timeout 300 python bench.py $X --steps 6 --warmup 2 --no-cpu-baseline --no-e2e --no-configs2 --extras-steps 4 2>$O/err.log | tail -1 > $O/b.json
  python - "shift $sh $X" $O/b.json <<'P' | tee -a $O/ab.log
import json,sys
name,p=sys.argv[1:3]
try:
    d=json.load(open(p)); r=lambda x:round(x,2); s=d.get("serial",{})
    print("%-28s"%name,"step",r(d["ms_per_step"]),"ms",r(d["value"]/1e6),"M/s | in-step chain",r(d["kernel_ms"]["k_chain"]),"| serial",r(s.get("ms_per_step",0)),"chain alone",r((s.get("aligned_kernel_ms") or {}).get("k_chain",0)))
except Exception as ex:
    print(name,"FAILED",ex)
P
done; done
