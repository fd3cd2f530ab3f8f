#!/bin/bash
# round 6, GPU call 7: record kernel v7 with (a) a & b & ~c as one v_bitop3, (b) the phase-ablation bits compiled out, (c) the <= 8 letters of an
# event as three aligned dword ORs instead of eight predicated byte stores: parity, then same-box A/B against the build before
cd "$(dirname "$0")/../.."
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r06g; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_transcriptome.py -m gpu -x -q 2>&1 | tail -4 ) | tee $O/pytest_parity.log
for name in ${VARIANTS:-base micro base micro}; do
  f=nanosim_amd/_variants/$name.so
  NANOSIM_AMD_LIB=$PWD/$f timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-e2e --extras-steps 3 2>$O/err_$name.log | tail -1 > $O/bench_$name.json
  python - $name $O/bench_$name.json <<'P' | tee -a $O/ab.log
import json,sys
name,p=sys.argv[1:3]
try:
    d=json.load(open(p))
except Exception as ex:
    print(name,"FAILED",ex); sys.exit(0)
r=lambda x:round(x,2)
s=d.get("serial",{}); e=d.get("errlog_on",{}); c=d.get("configs2",{}); f=d.get("chr1_fasta",{})
print(name,"step",r(d["ms_per_step"]),"ms",r(d["value"]/1e6),"M/s | aligned",{k:r(v) for k,v in d["kernel_ms"].items() if v>0.01},"frac",r(d["roofline"]["frac"]),
      "| serial",r(s.get("ms_per_step",0)),"al",{k:r(v) for k,v in (s.get("aligned_kernel_ms") or {}).items() if v>0.01},
      "| configs2",r(c.get("ms_per_step",0)),{k:r(v) for k,v in (c.get("aligned_batch",{}).get("kernel_ms") or {}).items() if v>0.01},
      "| chr1_fasta",r(f.get("ms_per_step",0)),{k:r(v) for k,v in (f.get("aligned_batch",{}).get("kernel_ms") or {}).items() if v>0.01}, r(f.get("roofline",{}).get("frac",0)))
P
done
