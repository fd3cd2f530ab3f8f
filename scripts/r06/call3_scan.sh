#!/bin/bash
# round 6, GPU call 3: the run scan of mutate_homo inside the first record pass of -k (hp_scan_chunks) against the scan kernel
# (NS_HP_SCAN_KERNEL=1 = the round-5 pipeline): parity first, then same-box A/B on configs[2].
cd "$(dirname "$0")/../.."
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r06c; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -8 ) | tee $O/pytest_parity.log
for rep in 1 2; do for v in fused kernel; do
  if [ $v = kernel ]; then export NS_HP_SCAN_KERNEL=1; else unset NS_HP_SCAN_KERNEL; fi
  timeout 300 python bench.py --genome chr1 --fastq --kmer-bias 5 --steps 5 --warmup 2 --no-cpu-baseline --no-e2e --no-extras 2>$O/err_$v.log | tail -1 > $O/bench_$v.json
  python - $v $O/bench_$v.json <<'P' | tee -a $O/ab.log
import json,sys
name,p=sys.argv[1:3]
try:
    d=json.load(open(p))
except Exception as ex:
    print(name,"FAILED",ex); sys.exit(0)
r=lambda x:round(x,2)
print(name,"step",r(d["ms_per_step"]),"ms",r(d["value"]/1e6),"M/s | aligned",r(d["aligned_batch"]["device_ms"]),{k:r(v) for k,v in d["kernel_ms"].items() if v>0.01},"frac",r(d["roofline"]["frac"]))
P
done; done
unset NS_HP_SCAN_KERNEL
( timeout 1500 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_parity.py 2>&1 | tail -8 ) | tee $O/pytest_rest.log
