#!/bin/bash
# rocprofv3 evidence for the bench lines of a round: for every configuration (ecoli_fasta = the default bench command, chr1_fasta,
# chr1_fastq_k5 = BASELINE configs[2], ecoli_fastq = configs[1] with --fastq) the bench line, the same command under --kernel-trace --stats, and separate PMC passes
# (FETCH_SIZE / WRITE_SIZE / SQ counters; never together with the trace domains) on a 200k-read aligned launch.
# Run on the GPU box through gpurun; outputs go to gpurun_out/<tag>/; afterwards, here: python scripts/summarise_pmc.py <tag>.
TAG=${1:-r04}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG
mkdir -p $O; cd /tmp; export TMPDIR=/tmp; ulimit -c 0
run_cfg() {
  key=$1; shift; ARGS="$@"
  [ -n "$SKIP_BENCH" ] || timeout 900 python $R/bench.py $EXTRA $ARGS > $O/bench_$key.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$key -o s -- python $R/bench.py --no-cpu-baseline --no-e2e --no-configs2 $ARGS > $O/bench_${key}_under_rocprof.log 2>&1
  B="python $R/bench.py --steps 1 --warmup 1 --reads 200000 --no-cpu-baseline --no-e2e --no-configs2 --aligned-only $ARGS"
  timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_${key}/fetch -o p -- $B > /dev/null 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_${key}/write -o p -- $B > /dev/null 2>&1
  timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/pmc_${key}/sq1 -o p -- $B > /dev/null 2>&1
  timeout 300 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM --kernel-trace --output-format csv -d $O/pmc_${key}/sq2 -o p -- $B > /dev/null 2>&1
  [ -n "$SKIP_BENCH" ] || grep -h "^{" $O/bench_$key.log | tail -1 | cut -c1-200
}
# SKIP_BENCH=1: only the rocprofv3 passes (the plain bench line of the configuration exists already)
CFGS=${CFGS:-"ecoli_fasta chr1_fasta chr1_fastq_k5 ecoli_fastq"}      # CFGS="ecoli_fastq" bash scripts/profile_round.sh r03: one configuration only
want() { case " $CFGS " in *" $1 "*) return 0;; esac; return 1; }
EXTRA="${EXTRA_FIRST:-}"       # the default command: the line the driver records (with its e2e legs and its configs2 object); EXTRA_FIRST="--no-e2e" skips the /dev/shm legs
want ecoli_fasta && run_cfg ecoli_fasta
EXTRA="--no-e2e --no-configs2"
want chr1_fasta && run_cfg chr1_fasta --genome chr1
want chr1_fastq_k5 && run_cfg chr1_fastq_k5 --genome chr1 --fastq --kmer-bias 5
want ecoli_fastq && run_cfg ecoli_fastq --fastq
want ecoli_fasta_errlog && run_cfg ecoli_fasta_errlog --errlog        # k_errlen + k_errlog: what every worker call of the CLI runs
