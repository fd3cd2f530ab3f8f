#!/bin/bash
cd "$(dirname "$0")/../.."
export NS_UCOOP_SHIFT=0 NS_STEP_GATE=0
O=gpurun_out/r06ar; mkdir -p $O
scripts/r06/call30_timeline.sh > $O/timeline_shift0.log 2>&1
grep -vE "rocprim|rp::|copyBuffer|fillBuffer" $O/timeline_shift0.log | tail -30
grep -E "copyBuffer|fillBuffer|rocprim:scan" $O/timeline_shift0.log | awk '$1>2.0' | head -30
