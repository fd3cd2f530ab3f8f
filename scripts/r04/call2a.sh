#!/bin/bash
# round 4, GPU call 2a: background-context A/B and the record / chain kernel variants (aligned-only A/B runs)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04b; mkdir -p $O; cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== A/B variants (aligned-only, configs[1])"; bash scripts/ab_run.sh --no-e2e --no-configs2 --no-extras 2>&1 | tee $O/ab_variants.log
( timeout 300 python scripts/r04/sweep_bg.py 6 2>&1 | grep -v "^\[" ) > $O/sweep_bg.log 2>&1; cat $O/sweep_bg.log
echo "== A/B FASTQ -k5 chr1"; VARIANTS="base wc7" bash scripts/ab_run.sh --no-e2e --no-configs2 --no-extras --genome chr1 --fastq --kmer-bias 5 2>&1 | tee $O/ab_variants_k5.log
( timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -5 ) > $O/pytest_parity.log 2>&1; cat $O/pytest_parity.log
