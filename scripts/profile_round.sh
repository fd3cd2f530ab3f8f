#!/bin/bash
# rocprofv3 evidence for the bench line: kernel-trace stats of the default bench command + separate PMC passes
# (FETCH_SIZE / WRITE_SIZE / SQ counters) on a 200k-read launch.  Run on the GPU box through gpurun; outputs go to
# gpurun_out/<tag>/; afterwards, here: python scripts/summarise_pmc.py <tag>; cp the stats csv + bench lines into profiles/<tag>/.
TAG=${1:-r01}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG
mkdir -p $O; cd /tmp; export TMPDIR=/tmp; ulimit -c 0
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-genome-run > $O/bench_under_rocprof.log 2>&1
timeout 300 python $R/bench.py --steps 5 --warmup 2 > $O/bench.log 2>&1
B="python $R/bench.py --steps 1 --warmup 1 --reads 200000 --no-cpu-baseline"
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o p -- $B > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o p -- $B > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/pmc_sq1 -o p -- $B > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM --kernel-trace --output-format csv -d $O/pmc_sq2 -o p -- $B > /dev/null 2>&1
grep -h "^{" $O/bench.log | tail -1
