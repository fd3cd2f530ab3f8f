#!/bin/bash
# round 6, GPU call 67 (call 65 + staged event stores for the reads of several pieces): chimeric batches — the longest reads of several pieces on the wave-per-read list too (NS_COOP_MULTI_SHIFT): parity, then the share
cd "$(dirname "$0")/../.."
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r06bw; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
( timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_metagenome.py tests/test_gpu_transcriptome.py tests/test_gpu_cli.py -m gpu -x -q 2>&1 | tail -3 ) | tee $O/pytest.log
run() {
  timeout 400 python bench.py --genome $1 $2 --steps 6 --warmup 2 --no-cpu-baseline --no-e2e --no-configs2 --extras-steps 3 2>$O/err.log | tail -1 > $O/b.json
  python - "$1 $2 shift=${NS_COOP_MULTI_SHIFT:-def}" $O/b.json <<'P' | tee -a $O/ab.log
import json,sys
name,p=sys.argv[1:3]
d=json.load(open(p)); r=lambda x:round(x,3); a=d["aligned_batch"]; k=a["kernel_ms"]; s=d.get("serial",{})
print("%-32s"%name,"step",r(d["ms_per_step"]),"ms",r(d["value"]/1e6),"M/s | aligned call",r(a["device_ms"]),"plan",r(k["plan(k_nseg+k_lengths+scan+sort)"]),"chain",r(k["k_chain"]),"record",r(k["k_materialise"]),"| chain alone",r((s.get("aligned_kernel_ms") or {}).get("k_chain",0)))
P
}
run ecoli ""
for sh in def 3 7 8 10; do
  if [ $sh = def ]; then unset NS_COOP_MULTI_SHIFT; else export NS_COOP_MULTI_SHIFT=$sh; fi
  run chr1 --chimeric
done
unset NS_COOP_MULTI_SHIFT
run grch38 --chimeric
