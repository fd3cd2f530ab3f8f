#!/bin/bash
# -hp -k 5 (+FASTQ) path: kernel-trace stats and per-read SQ / HBM counters of the homopolymer kernels
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/hp; mkdir -p $O; cd /tmp; export TMPDIR=/tmp; ulimit -c 0
ARGS="--kmer-bias 5 $1"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline $ARGS > $O/bench.log 2>&1
head -12 $O/stats/*/s_kernel_stats.csv 2>/dev/null | cut -c1-160 || find $O/stats -name "*kernel_stats.csv" | head -1 | xargs head -12 | cut -c1-160
timeout 200 bash $R/scripts/pmc_kernels.sh "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" 0 "$ARGS"
timeout 200 bash $R/scripts/pmc_kernels.sh "FETCH_SIZE" 0 "$ARGS"
timeout 200 bash $R/scripts/pmc_kernels.sh "WRITE_SIZE" 0 "$ARGS"
tail -1 $O/bench.log | cut -c1-900
