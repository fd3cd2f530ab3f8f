#!/bin/bash
# round 5, GPU call 3: where the time of the adopted kernels goes — NS_DEBUG_SKIP ablation of the record kernel (aligned batch alone),
# then the rocprofv3 passes (kernel stats + PMC, separate passes) for configs[1], configs[1] with the error profile, configs[2].
cd "$(dirname "$0")/../.."
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r05c; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
# bits (ns_materialise.h): 1 no sub-run loads / no final pass, 2 no letters, 16 no 16-byte stores, 32 no merge, 64 no letter-word Philox, 8192 no chunk descriptors
for d in 0 1 2 16 32 64 8192 34 99 8291 0; do echo -n "skip=$d "; NS_DEBUG_SKIP=$d timeout 90 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --aligned-only 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('k_materialise', round(d['kernel_ms']['k_materialise'],3), 'k_chain', round(d['kernel_ms']['k_chain'],3), 'aligned batch', round(d['aligned_batch']['device_ms'],3))"; done 2>&1 | tee $O/ablate_materialise.log
SKIP_BENCH=1 CFGS="ecoli_fasta ecoli_fasta_errlog chr1_fastq_k5" bash scripts/profile_round.sh r05c 2>&1 | tail -5
ls -R $O | head -50
