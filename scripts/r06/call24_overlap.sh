#!/bin/bash
# round 6, GPU call 24: probe — two engine contexts side by side at several lags: is there time to win by running the chain of one
# (half-)batch under the record kernel of another?
cd "$(dirname "$0")/../.."
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r06aa; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python scripts/r06/overlap_probe.py 2>$O/err.log | tee $O/probe.log
tail -5 $O/err.log
