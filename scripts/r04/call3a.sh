#!/bin/bash
# round 4, GPU call 3a: rocprofv3 evidence of the headline configuration at HEAD (kernel stats + PMC passes), bench line without the
# heavy /dev/shm legs
R=$GRAFT_REPO_ROOT; cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
CFGS="ecoli_fasta" EXTRA_FIRST="--no-e2e" bash scripts/profile_round.sh r04 2>&1 | tail -5
ls gpurun_out/r04
