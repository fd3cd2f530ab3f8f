#!/bin/bash
# round 6, GPU call 44: parity of the default build (three loop iterations per lane in the wave-per-read unaligned chain)
cd "$(dirname "$0")/../.."
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r06ax; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
( timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_metagenome.py tests/test_gpu_transcriptome.py -m gpu -x -q 2>&1 | tail -4 ) | tee $O/pytest.log
( NS_SWEEP_SEED=0xBEEF01 timeout 1500 python scripts/parity_sweep.py 40000 2>&1 | grep -i "unaligned" ) | tee $O/parity_sweep_unaligned.log
