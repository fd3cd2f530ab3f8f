#!/bin/bash
# round 6, GPU call 79: configs[2] at its full read count on the last build (after the chimeric chain work): the hash of the record stream must equal
# the one of call 53 and of round 5 (f57d9f72aed9fba1 / e85cbbf8d72e39d1)
cd "$(dirname "$0")/../.."
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r06fs2; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
s=$(date +%s)
timeout 1500 python scripts/fullsize_stream.py --config 2 > $O/fullsize_stream_configs2.log 2>$O/err_2.log
echo "config 2: rc $? in $(( $(date +%s) - s )) s"; tail -1 $O/fullsize_stream_configs2.log | cut -c1-700
