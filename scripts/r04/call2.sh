#!/bin/bash
# round 4, second GPU call: transcriptome block walk on the GPU, background-context A/B, configs[3] / configs[4] bench lines, CLI lockstep
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04b; mkdir -p $O; cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
( timeout 900 python -m pytest tests/test_gpu_transcriptome.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -25 ) > $O/pytest_trx_parity.log 2>&1
( timeout 600 python -m pytest tests/test_gpu_cli.py -m gpu -x -q -k "transcriptome or two_context" 2>&1 | tail -25 ) > $O/pytest_cli_trx.log 2>&1
( timeout 600 python scripts/parity_trx_big.py 20000 2>&1 | tail -8 ) > $O/parity_trx_big.log 2>&1
( timeout 300 python scripts/r04/sweep_bg.py 6 2>&1 | grep -v "^\[" ) > $O/sweep_bg.log 2>&1
timeout 300 python bench.py --no-cpu-baseline --no-e2e --no-configs2 > $O/bench_ecoli_noextra.json 2> $O/bench_ecoli_noextra.err
( timeout 300 python scripts/bench_transcriptome.py; timeout 300 python scripts/bench_transcriptome.py --model-ir ) > $O/bench_transcriptome.log 2>&1
( NS_CLI_TRACE=1 timeout 300 python scripts/bench_cli.py -n 3000000 ) > $O/bench_cli_default.log 2>&1
timeout 600 python bench.py --metagenome --no-e2e > $O/bench_zymo10_metagenome.json 2> $O/bench_zymo10.err
timeout 900 python bench.py --genome grch38 --chimeric --no-e2e --cpu-sample 2000 > $O/bench_grch38_chimeric.json 2> $O/bench_grch38.err
for f in pytest_trx_parity pytest_cli_trx parity_trx_big sweep_bg bench_transcriptome; do echo "== $f"; tail -12 $O/$f.log; done
tail -4 $O/bench_cli_default.log
for f in bench_ecoli_noextra bench_zymo10_metagenome bench_grch38_chimeric; do echo "== $f"; cut -c1-260 $O/$f.json; tail -2 $O/${f%%_*}*.err 2>/dev/null | head -3; done
echo "== A/B record kernel variants (aligned-only, configs[1])"; bash scripts/ab_run.sh --no-e2e --no-configs2 --no-extras 2>&1 | tee $O/ab_wordcache.log
echo "== A/B FASTQ -k5 chr1"; VARIANTS="base wc7" bash scripts/ab_run.sh --no-e2e --no-configs2 --no-extras --genome chr1 --fastq --kmer-bias 5 2>&1 | tee $O/ab_wordcache_k5.log
