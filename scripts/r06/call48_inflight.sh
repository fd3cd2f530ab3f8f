#!/bin/bash
# round 6, GPU call 48: the `two_steps_in_flight` object of the bench line
cd "$(dirname "$0")/../.."
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r06bc; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
for rep in 1 2; do
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-e2e --no-configs2 --extras-steps 6 2>$O/err.log | tail -1 > $O/b.json
python - $O/b.json <<'P' | tee -a $O/inflight.log
import json,sys
d=json.load(open(sys.argv[1])); r=lambda x:round(x,3)
print("step", r(d["ms_per_step"]), "ms", r(d["value"]/1e6), "M reads/s frac", r(d["roofline"]["frac"]), "| serial", r(d["serial"]["ms_per_step"]), "| two_steps_in_flight", d.get("two_steps_in_flight"))
P
done
tail -3 $O/err.log
