#!/bin/bash
# round 6, GPU call 59: a batch whose length window rejects every other read per pass: the retry list's counter took one atomic per rejected read
# (base = the commit before) against one per wavefront; parity of the retry paths
cd "$(dirname "$0")/../.."
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r06bo; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
( timeout 1800 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3 ) | tee $O/pytest.log
echo "== base"; NANOSIM_AMD_LIB=$PWD/nanosim_amd/_variants/base.so timeout 300 python scripts/r06/narrow_probe.py 2>/dev/null | tail -3
echo "== new"; timeout 300 python scripts/r06/narrow_probe.py 2>/dev/null | tail -3
