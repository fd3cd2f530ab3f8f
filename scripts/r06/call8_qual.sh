#!/bin/bash
# round 6, GPU call 8: k_qualities with its halfword pairs built by v_perm and the bucket arithmetic as packed 16-bit add / shift: parity, then A/B (FASTQ, configs[1] genome)
cd "$(dirname "$0")/../.."
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r06h; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_transcriptome.py tests/test_gpu_metagenome.py -m gpu -x -q 2>&1 | tail -4 ) | tee $O/pytest_parity.log
for name in micro qpk micro qpk; do
  f=nanosim_amd/_variants/$name.so
  NANOSIM_AMD_LIB=$PWD/$f timeout 300 python bench.py --fastq --aligned-only --steps 6 --warmup 2 --no-cpu-baseline --no-e2e --no-extras 2>$O/err_$name.log | tail -1 > $O/bench_$name.json
  NANOSIM_AMD_LIB=$PWD/$f timeout 300 python bench.py --genome chr1 --fastq --kmer-bias 5 --aligned-only --steps 4 --warmup 1 --no-cpu-baseline --no-e2e --no-extras 2>$O/err2_$name.log | tail -1 > $O/bench2_$name.json
  python - $name $O/bench_$name.json $O/bench2_$name.json <<'P' | tee -a $O/ab.log
import json,sys
name,p,p2=sys.argv[1:4]
r=lambda x:round(x,2)
for tag,f in (("ecoli fastq",p),("chr1 fastq -k 5",p2)):
    try:
        d=json.load(open(f))
        print(name,tag,"step",r(d["ms_per_step"]),"ms |",{k:r(v) for k,v in d["kernel_ms"].items() if v>0.01})
    except Exception as ex:
        print(name,tag,"FAILED",ex)
P
done
