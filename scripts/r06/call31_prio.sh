#!/bin/bash
# round 6, GPU call 31: the step companion's streams at high priority (own pool of hardware queues) + the retry passes' k_lengths within 72 VGPRs:
# parity of the touched paths, the step A/B, the timeline again
cd "$(dirname "$0")/../.."
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r06ah; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -4 ) | tee $O/pytest.log
for rep in 1 2; do for cfg in "0 31" "1 31" "1 6" "1 8"; do
  set -- $cfg
  NS_STEP_PRIO=$1 NS_UWIDE_SHIFT=$2 timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-e2e --no-configs2 --no-extras 2>$O/err.log | tail -1 > $O/b.json
  python - "prio=$1 uwide_shift=$2" $O/b.json <<'P' | tee -a $O/ab.log
import json,sys
name,p=sys.argv[1:3]
try:
    d=json.load(open(p)); r=lambda x:round(x,2); u=d["unaligned_batch"]; a=d["aligned_batch"]
    print("%-24s"%name,"step",r(d["ms_per_step"]),"ms",r(d["value"]/1e6),"M/s | aligned call",r(a["device_ms"]),"chain",r(a["kernel_ms"]["k_chain"]),"record",r(a["kernel_ms"]["k_materialise"]),
          "| unaligned call",r(u["device_ms"]),"chain",r(u["kernel_ms"]["k_chain"]),"dense",r(u["kernel_ms"]["k_materialise"]))
except Exception as ex:
    print(name,"FAILED",ex)
P
done; done
NS_UWIDE_SHIFT=31 scripts/r06/call30_timeline.sh > $O/timeline_prio.log 2>&1
grep -E "k_chain|k_lengths|k_materialise|k_names|dense" $O/timeline_prio.log
