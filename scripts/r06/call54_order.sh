#!/bin/bash
# round 6, GPU call 54: the visiting order of a batch dealt into bins of the key (k_order_*) against the full sort (NS_EXACT_ORDER=1): parity, alternating bench lines
cd "$(dirname "$0")/../.."
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r06bj; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
( timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cli.py tests/test_gpu_sinks.py -m gpu -x -q 2>&1 | tail -3 ) | tee $O/pytest.log
for rep in 1 2 3 4; do for ex in 1 0; do
  if [ $ex = 1 ]; then export NS_EXACT_ORDER=1; else unset NS_EXACT_ORDER; fi
  timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-e2e --no-configs2 --extras-steps 4 2>$O/err.log | tail -1 > $O/b.json
  python - "exact_sort=$ex" $O/b.json <<'P' | tee -a $O/ab.log
import json,sys
name,p=sys.argv[1:3]
d=json.load(open(p)); r=lambda x:round(x,3); a=d["aligned_batch"]; k=a["kernel_ms"]; u=d["unaligned_batch"]; s=d.get("serial",{})
print("%-14s"%name,"step",r(d["ms_per_step"]),"ms",r(d["value"]/1e6),"M/s frac",r(d["roofline"]["frac"]),"| aligned call",r(a["device_ms"]),"plan",r(k["plan(k_nseg+k_lengths+scan+sort)"]),"chain",r(k["k_chain"]),"record",r(d["roofline"]["kernel_ms"]),"| unaligned plan",r(u["kernel_ms"]["plan(k_nseg+k_lengths+scan+sort)"]),"| serial",r(s.get("ms_per_step",0)),"chain alone",r((s.get("aligned_kernel_ms") or {}).get("k_chain",0)))
P
done; done
