#!/bin/bash
# round 6, GPU call 2: v7 at other tile sizes / occupancies (same-box A/B), then the new recovery-path tests (NS_CAP_RATE_SCALE, NS_HP_CAP_SHIFT)
# and the multi-rank CLI tests touched by --merge.
cd "$(dirname "$0")/../.."
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r06b; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
for name in ${VARIANTS:-v7 w8 tc1w8 tc4w6 v7}; do
  f=nanosim_amd/_variants/$name.so
  NANOSIM_AMD_LIB=$PWD/$f timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-e2e --no-extras 2>$O/err_$name.log | tail -1 > $O/bench_$name.json
  python - $name $O/bench_$name.json <<'P' | tee -a $O/ab.log
import json,sys
name,p=sys.argv[1:3]
try:
    d=json.load(open(p))
except Exception as ex:
    print(name,"FAILED",ex); sys.exit(0)
r=lambda x:round(x,2)
c=d.get("configs2",{}); f=d.get("chr1_fasta",{})
print(name,"step",r(d["ms_per_step"]),"ms",r(d["value"]/1e6),"M/s | aligned",{k:r(v) for k,v in d["kernel_ms"].items() if v>0.01},"frac",r(d["roofline"]["frac"]),
      "| configs2",r(c.get("ms_per_step",0)),{k:r(v) for k,v in (c.get("aligned_batch",{}).get("kernel_ms") or {}).items() if v>0.01},
      "| chr1_fasta",r(f.get("ms_per_step",0)),{k:r(v) for k,v in (f.get("aligned_batch",{}).get("kernel_ms") or {}).items() if v>0.01}, "frac", r(f.get("roofline",{}).get("frac",0)))
P
done
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_metagenome.py tests/test_gpu_transcriptome.py -m gpu -x -q -k "overflow or capacity" 2>&1 | tail -8
  timeout 900 python -m pytest tests/test_gpu_cli.py tests/test_gpu_sinks.py -m gpu -x -q -k "ranks or sub_files" 2>&1 | tail -8 ) | tee $O/pytest_new.log
