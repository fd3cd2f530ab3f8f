#!/bin/bash
# round 5, GPU call 22: counters of the UNALIGNED worker call's kernels (the wave-per-read chain k_chain<true, true> on its LDS image and the
# dense record kernel) — one serial step of 10^6 reads (950 000 aligned, then 50 000 unaligned on the same context: every kernel alone),
# separate rocprofv3 PMC passes as scripts/profile_round.sh makes them
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05y; mkdir -p $O; cd /tmp; export TMPDIR=/tmp; ulimit -c 0
export HSA_ENABLE_IPC_MODE_LEGACY=0
B="python $R/bench.py --steps 1 --warmup 1 --serial --no-cpu-baseline --no-e2e --no-configs2 --no-extras"
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_unaligned/fetch -o p -- $B > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_unaligned/write -o p -- $B > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/pmc_unaligned/sq1 -o p -- $B > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM --kernel-trace --output-format csv -d $O/pmc_unaligned/sq2 -o p -- $B > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_unaligned -o s -- $B > $O/bench_serial_under_rocprof.log 2>&1
ls $O/pmc_unaligned/*/ | head
