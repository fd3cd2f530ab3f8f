#!/bin/bash
# round 5, GPU call 5: k_errlog and the dense record kernel with aligned LDS accesses only — parity, then the default line's objects
cd "$(dirname "$0")/../.."
O=gpurun_out/r05f; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
( timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_metagenome.py -m gpu -x -q 2>&1 | tail -5 ) | tee $O/pytest_parity.log
timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-e2e --no-configs2 2>$O/bench.err | tail -1 > $O/bench_ecoli_fasta.json
python - $O/bench_ecoli_fasta.json <<'P' | tee $O/bench_summary.log
import json,sys
d=json.load(open(sys.argv[1])); r=lambda x:round(x,3)
print("step", r(d["ms_per_step"]), "ms", r(d["value"]/1e6), "M reads/s | aligned", {k:r(v) for k,v in d["kernel_ms"].items() if v>0.01}, "| unaligned", {k:r(v) for k,v in d["unaligned_batch"]["kernel_ms"].items() if v>0.01})
for k in ("serial","errlog_on"): print(k, {a:(r(b) if isinstance(b,float) else b) for a,b in d[k].items() if not isinstance(b,(str,))})
P
