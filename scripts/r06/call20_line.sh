#!/bin/bash
# round 6: the driver's bench command (twice) on the current build
cd "$(dirname "$0")/../.."
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r06x; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
for i in 1 2; do
S=$(date +%s); timeout 600 python bench.py 2>$O/bench_default_$i.err | tail -1 > $O/bench_default_$i.json; echo "wall $(( $(date +%s) - S )) s"
python - $O/bench_default_$i.json <<'P' | tee -a $O/bench_summary.log
import json,sys
d=json.load(open(sys.argv[1])); r=lambda x:round(x,3)
print("step", r(d["ms_per_step"]), "ms", r(d["value"]/1e6), "M reads/s frac", r(d["roofline"]["frac"]), {k:r(v) for k,v in d["kernel_ms"].items() if v>0.01})
for key in ("configs2","chr1_fasta","trained_shape"):
    c=d.get(key,{}); print(" ", key, r(c.get("ms_per_step",0)), r(c.get("value",0)/1e6), "M reads/s", {k:r(v) for k,v in (c.get("aligned_batch",{}).get("kernel_ms") or {}).items() if v>0.01}, "frac", r(c.get("roofline",{}).get("frac",0)))
P
done
