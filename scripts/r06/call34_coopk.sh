#!/bin/bash
# round 6, GPU call 34: the wave-per-read unaligned chain with four iterations per lane: parity, the unaligned call alone, the step
cd "$(dirname "$0")/../.."
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r06ak; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
export NS_UWIDE_SHIFT=31 NS_STEP_PRIO=0
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -4 ) | tee $O/pytest.log
for k in 1 4; do echo "== NS_UCOOP_K=$k (alone, foreground: all reads on the wave-per-read chain)"; NS_UCOOP_K=$k K=3 timeout 200 python scripts/r06/unaligned_probe.py 2>/dev/null | tail -2; done | tee $O/alone.log
for rep in 1 2; do for cfg in "1 3" "4 3" "4 2" "4 1" "4 0"; do
  set -- $cfg
  NS_UCOOP_K=$1 NS_UCOOP_SHIFT=$2 timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-e2e --no-configs2 --no-extras 2>$O/err.log | tail -1 > $O/b.json
  python - "K=$1 ucoop_shift=$2" $O/b.json <<'P' | tee -a $O/ab.log
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
