#!/bin/bash
# round 6, GPU call 68: per-wavefront durations of the thread-per-read chain in a chimeric batch (-DNS_CHAIN_CLOCK build: build_ab/libns_clock.so),
# wavefronts with a read of several pieces against the others, at several shares of those reads on the wave-per-read list
cd "$(dirname "$0")/../.."
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r06by; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0 NANOSIM_AMD_LIB=$PWD/build_ab/libns_clock.so
for cfg in "chr1 --chimeric def" "chr1 --chimeric 3"; do
  set -- $cfg
  if [ $3 = def ]; then unset NS_COOP_MULTI_SHIFT; else export NS_COOP_MULTI_SHIFT=$3; fi
  fl=$2; [ $fl = - ] && fl=
  echo "== $1 $fl shift=$3" | tee -a $O/clock.log
  timeout 400 python bench.py --genome $1 $fl --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-configs2 --no-extras 2>$O/err.log >/dev/null
  grep "chain clock" $O/err.log | tail -4 | tee -a $O/clock.log
done
