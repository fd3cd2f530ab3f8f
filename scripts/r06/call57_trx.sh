#!/bin/bash
# round 6, GPU call 57: k_trx_commit's counters spread over 64 copies; the second stream (k_names) at high priority (NS_STREAM2_PRIO=1): parity, benches
cd "$(dirname "$0")/../.."
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r06bm; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
( timeout 1800 python -m pytest tests/test_gpu_transcriptome.py tests/test_gpu_cli.py -m gpu -x -q 2>&1 | tail -3 ) | tee $O/pytest.log
for pr in 0 1 0 1; do
  echo "stream2_prio=$pr trx $(NS_STREAM2_PRIO=$pr timeout 300 python scripts/bench_transcriptome.py 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["ms_per_batch"],3), "ms", round(d["value"]/1e6,1), "M reads/s", {k:round(v,3) for k,v in d["kernel_ms"].items() if v>0.01})')" | tee -a $O/trx.log
done
for rep in 1 2 3; do for pr in 0 1; do
  NS_STREAM2_PRIO=$pr timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-e2e --no-configs2 --no-extras 2>$O/err.log | tail -1 > $O/b.json
  python - "stream2_prio=$pr" $O/b.json <<'P' | tee -a $O/ab.log
import json,sys
name,p=sys.argv[1:3]
d=json.load(open(p)); r=lambda x:round(x,3); a=d["aligned_batch"]; k=a["kernel_ms"]
print("%-16s"%name,"step",r(d["ms_per_step"]),"ms",r(d["value"]/1e6),"M/s frac",r(d["roofline"]["frac"]),"| aligned call",r(a["device_ms"]),"chain",r(k["k_chain"]),"record stage",r(k["k_materialise"]),"kernel",r(d["roofline"]["kernel_ms"]))
P
done; done
