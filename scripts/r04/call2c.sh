#!/bin/bash
# round 4, GPU call 2c: bench lines of configs[1] (with serial / errlog_on), configs[4], configs[3]; the CLI in lockstep
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04d; mkdir -p $O; cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python bench.py --no-cpu-baseline --no-e2e --no-configs2 > $O/bench_ecoli_noextra.json 2> $O/bench_ecoli_noextra.err
( NS_CLI_TRACE=1 timeout 300 python scripts/bench_cli.py -n 3000000 ) > $O/bench_cli_default.log 2>&1
timeout 600 python bench.py --metagenome --no-e2e > $O/bench_zymo10_metagenome.json 2> $O/bench_zymo10.err
timeout 900 python bench.py --genome grch38 --chimeric --no-e2e --cpu-sample 2000 > $O/bench_grch38_chimeric.json 2> $O/bench_grch38.err
tail -6 $O/bench_cli_default.log
for f in bench_ecoli_noextra bench_zymo10_metagenome bench_grch38_chimeric; do echo "== $f"; cut -c1-260 $O/$f.json; done; tail -3 $O/*.err
