#!/bin/bash
# round 6, GPU call 55: binned visiting order also for the transcriptome candidates; parity of genome / metagenome / transcriptome + CLI, transcriptome bench
cd "$(dirname "$0")/../.."
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r06bk; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
( timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_metagenome.py tests/test_gpu_transcriptome.py tests/test_gpu_cli.py -m gpu -x -q 2>&1 | tail -3 ) | tee $O/pytest.log
for ex in 1 0 1 0; do
  if [ $ex = 1 ]; then export NS_EXACT_ORDER=1; else unset NS_EXACT_ORDER; fi
  echo "exact_sort=$ex $(timeout 300 python scripts/bench_transcriptome.py 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["ms_per_batch"],3), "ms", round(d["value"]/1e6,1), "M reads/s", {k:round(v,3) for k,v in d["kernel_ms"].items() if v>0.01})')" | tee -a $O/trx.log
done
