#!/bin/bash
# round 6, GPU call 22: the wave-per-read chain (the longest reads of a batch: what the chain phase of a call waits for) on the one-word segments of the
# full columns with a 65 536-cell guide, four columns' reads side by side, against the fp64 loop (one column after the other): parity, then A/B
cd "$(dirname "$0")/../.."
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r06y2; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_metagenome.py tests/test_gpu_transcriptome.py -m gpu -x -q 2>&1 | tail -4 ) | tee $O/pytest_parity.log
for rep in 1 2; do for name in base coop2; do for X in "" "--trained-shape"; do
  f=nanosim_amd/_variants/$name.so
  NANOSIM_AMD_LIB=$PWD/$f timeout 300 python bench.py $X --steps 6 --warmup 2 --no-cpu-baseline --no-e2e --no-configs2 --extras-steps 4 2>$O/err.log | tail -1 > $O/b.json
  python - "$name$X" $O/b.json <<'P' | tee -a $O/ab.log
import json,sys
name,p=sys.argv[1:3]
try:
    d=json.load(open(p)); r=lambda x:round(x,2); s=d.get("serial",{})
    print("%-26s"%name,"step",r(d["ms_per_step"]),"ms",r(d["value"]/1e6),"M/s | in-step chain",r(d["kernel_ms"]["k_chain"]),"record",r(d["kernel_ms"]["k_materialise"]),"| serial",r(s.get("ms_per_step",0)),"chain alone",r((s.get("aligned_kernel_ms") or {}).get("k_chain",0)))
except Exception as ex:
    print(name,"FAILED",ex)
P
done; done; done
