#!/bin/bash
# round 6, GPU call 27: the unaligned worker call alone under rocprofv3: the wave-per-read and the workgroup-per-read chain kernels one by one
cd "$(dirname "$0")/../.."
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
O=$PWD/gpurun_out/r06ad; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
( timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "workgroup_per_read or background_context" 2>&1 | tail -4 ) | tee $O/pytest_uwide.log
R=$PWD
cd /tmp && export TMPDIR=/tmp
for sh in 31 6 4; do
  rm -rf /tmp/prof_$sh
  NS_UWIDE_SHIFT=$sh timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$sh -- python $R/scripts/r06/unaligned_probe.py > $O/probe_$sh.log 2>$O/err_$sh.log
  echo "== NS_UWIDE_SHIFT=$sh" | tee -a $O/kernels.log
  tail -3 $O/probe_$sh.log | tee -a $O/kernels.log
  f=$(find /tmp/prof_$sh -name "*kernel_stats.csv" | head -1)
  head -8 "$f" | cut -c1-260 | tee -a $O/kernels.log
done
