#!/bin/bash
# round 6, GPU call 53: BASELINE's configurations at their full read counts on the last build (scripts/fullsize_stream.py): the hashes of the
# record streams must equal each other (one worker against 8 index ranges) AND those of round 5 (profiles/r05/fullsize_stream_configs*.log)
cd "$(dirname "$0")/../.."
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r06bi; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
for c in ${CONFIGS_TO_RUN:-1 2 3}; do
  s=$(date +%s)
  timeout 3000 python scripts/fullsize_stream.py --config $c > $O/fullsize_stream_configs$c.log 2>$O/err_$c.log
  echo "config $c: rc $? in $(( $(date +%s) - s )) s"; tail -1 $O/fullsize_stream_configs$c.log | cut -c1-400
done
