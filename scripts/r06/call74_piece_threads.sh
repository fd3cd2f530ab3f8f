#!/bin/bash
# round 6, GPU call 74: chimeric batches — the reads of several pieces on the thread-per-read side as a thread per PIECE (k_chain piece modes 1 + 2, third stream);
# NS_NO_PIECE_THREADS=1: one thread per read (the build before).  Parity first (the 131 072-read batch against batches of 4 096 + oracle), then same-box A/B.
cd "$(dirname "$0")/../.."
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r06pt; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
( timeout 1800 python -m pytest tests/test_gpu_fullsize_refs.py tests/test_gpu_fullsize.py tests/test_gpu_parity.py -m gpu -x -q -k "chimeric or both_chain_lists or full_size" 2>&1 | tail -4 ) | tee $O/pytest.log
run() {
  timeout 400 python bench.py --genome $1 $2 --steps 6 --warmup 2 --no-cpu-baseline --no-e2e --no-configs2 --extras-steps 3 2>$O/err.log | tail -1 > $O/b.json
  python - "$1 $2 shift=${NS_COOP_MULTI_SHIFT:-def} one_thread_per_read=${NS_NO_PIECE_THREADS:-0}" $O/b.json <<'P' | tee -a $O/ab.log
import json,sys
name,p=sys.argv[1:3]
d=json.load(open(p)); r=lambda x:round(x,3); a=d["aligned_batch"]; k=a["kernel_ms"]; s=d.get("serial",{})
print("%-58s"%name,"step",r(d["ms_per_step"]),"ms",r(d["value"]/1e6),"M/s | aligned call",r(a["device_ms"]),"plan",r(k["plan(k_nseg+k_lengths+scan+sort)"]),"chain",r(k["k_chain"]),"record",r(k["k_materialise"]),"| chain alone",r((s.get("aligned_kernel_ms") or {}).get("k_chain",0)))
P
}
NS_NO_PIECE_THREADS=1 run chr1 --chimeric
run chr1 --chimeric
NS_NO_PIECE_THREADS=1 run chr1 --chimeric
run chr1 --chimeric
for sh in 7 9 12; do NS_COOP_MULTI_SHIFT=$sh run chr1 --chimeric; done
run grch38 --chimeric
run ecoli ""
