#!/bin/bash
# round 6, GPU call 6: the rocprofv3 evidence behind the roofline numbers of the v7 record kernel (scripts/profile_round.sh: the bench
# command under --kernel-trace --stats, then separate PMC passes on a 200 000-read aligned launch), four configurations
cd "$(dirname "$0")/../.."
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
export HSA_ENABLE_IPC_MODE_LEGACY=0
CFGS="ecoli_fasta chr1_fasta chr1_fastq_k5 ecoli_fasta_errlog" bash scripts/profile_round.sh r06 2>&1 | tail -6
du -sh gpurun_out/r06 | tail -1
# keep the CSVs only (the merge back is capped at 64 MiB)
find gpurun_out/r06 -type f ! -name "*.csv" ! -name "*.log" ! -name "*.json" -delete
du -sh gpurun_out/r06 | tail -1
