#!/bin/bash
# round 6, GPU call 35: where the wave-per-read unaligned chain spends its 2.6 ms: without its error lists, without its event stores (NS_ABLATE build)
cd "$(dirname "$0")/../.."
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r06al; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
export NS_UWIDE_SHIFT=31 NANOSIM_AMD_LIB=$PWD/nanosim_amd/_variants/ablate.so
for k in 1 4; do for skip in 0 $((1<<20)) $((1<<21)); do
  echo "== NS_UCOOP_K=$k NS_DEBUG_SKIP=$skip"; NS_UCOOP_K=$k NS_DEBUG_SKIP=$skip K=2 timeout 200 python scripts/r06/unaligned_probe.py 2>&1 | tail -2 | cut -c1-200
done; done | tee $O/ablate.log
