#!/bin/bash
# round 6, GPU call 71: rocprofv3 evidence of the last build: the default bench command under --kernel-trace --stats (and its PMC passes), then a configs[3] step
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
R=$GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
CFGS="ecoli_fasta" EXTRA_FIRST="--no-e2e" bash $R/scripts/profile_round.sh r06p2
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r06p2/stats_grch38_chimeric -o s -- python $R/bench.py --genome grch38 --chimeric --no-cpu-baseline --no-e2e --no-configs2 --no-extras > $R/gpurun_out/r06p2/bench_grch38_chimeric_under_rocprof.log 2>&1
find $R/gpurun_out/r06p2 -name "*kernel_stats.csv" | head
