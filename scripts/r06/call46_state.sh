#!/bin/bash
# round 6, GPU call 46: the two states of the record kernel (4.7 / 5.0 ms): 16 processes, each prints its kernel time and buffer addresses
cd "$(dirname "$0")/../.."
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r06az; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0 PROBE_TMP=/tmp/probe_model
mkdir -p $PROBE_TMP
for i in $(seq 1 16); do timeout 120 python scripts/r06/record_state_probe.py 2>/dev/null | tail -1; done | tee $O/state.log
