#!/bin/bash
# round 5, GPU call 1: the prepared variants (scripts/ab_build.sh with the flags of round 4; the flags and their code are gone since: DESIGN.md section 6) timed FIRST — per variant one bench run whose line carries the
# two-context step, the serial step (each worker call's kernels alone), and errlog_on — then the bit-exact parity tests of each.
cd "$(dirname "$0")/../.."
O=gpurun_out/r05a; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
for name in ${VARIANTS:-base v2 v2m5 var8 bitop3 rekey errlog3 all base}; do
  f=nanosim_amd/_variants/$name.so
  NANOSIM_AMD_LIB=$PWD/$f timeout 200 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-e2e --no-configs2 --extras-steps 3 2>$O/err_$name.log | tail -1 > $O/bench_$name.json
  python - $name $O/bench_$name.json <<'P' | tee -a $O/ab.log
import json,sys
name,p=sys.argv[1:3]
try:
    d=json.load(open(p))
except Exception as ex:
    print(name,"FAILED",ex); sys.exit(0)
r=lambda x:round(x,2)
s=d.get("serial",{}); e=d.get("errlog_on",{})
print(name,"step",r(d["ms_per_step"]),"ms | in-step aligned",{k:r(v) for k,v in d["kernel_ms"].items() if v>0.01},"unaligned",{k:r(v) for k,v in d["unaligned_batch"]["kernel_ms"].items() if v>0.01},
      "| serial",r(s.get("ms_per_step",0)),"al",{k:r(v) for k,v in (s.get("aligned_kernel_ms") or {}).items() if v>0.01},"un",{k:r(v) for k,v in (s.get("unaligned_kernel_ms") or {}).items() if v>0.01},
      "| errlog_on",r(e.get("ms_per_step",0)),"k_errlog",r(e.get("k_errlog_ms",0)))
P
done
for name in ${PARITY:-v2 v2m5 var8 bitop3 rekey errlog3 all}; do
  f=nanosim_amd/_variants/$name.so
  echo "== $name parity"; ( NANOSIM_AMD_LIB=$PWD/$f timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3 ) | tee $O/parity_$name.log
done
( timeout 300 python -m pytest tests/test_gpu_zz_characterize.py -m gpu -q 2>&1 | tail -5 ) | tee $O/pytest_characterize.log
