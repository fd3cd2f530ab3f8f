#!/bin/bash
# round 5, GPU call 13: BASELINE configs[2] (10^7 reads, FASTQ, -hp -k 5) and configs[3] (10^8 reads, --chimeric) at their FULL read counts:
# the record stream of every read generated under two partitions of the read indices (one worker / 8 index ranges, different batch sizes)
# must be the same bytes (scripts/fullsize_stream.py: XXH3-64 of the stream, totals)
cd "$(dirname "$0")/../.."
O=gpurun_out/r05p; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 200 python scripts/fullsize_stream.py --config 1 2>$O/fullsize1.err | tee $O/fullsize_configs1.log | tail -1 | cut -c1-600
timeout 600 python scripts/fullsize_stream.py --config 2 2>$O/fullsize2.err | tee $O/fullsize_configs2.log | tail -1 | cut -c1-600
timeout 1200 python scripts/fullsize_stream.py --config 3 2>$O/fullsize3.err | tee $O/fullsize_configs3.log | tail -1 | cut -c1-600
tail -n 3 $O/fullsize1.err $O/fullsize2.err $O/fullsize3.err
