#!/bin/bash
# round 6, GPU call 17: trained-shape model — prefix length (NS_TAIL_BITS) x workgroup size of k_chain (NS_CHAIN_BLOCK), serial step (chain alone)
cd "$(dirname "$0")/../.."
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r06q; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
for tb in 12 11 10 9; do for cb in 256 512; do
  NS_TAIL_BITS=$tb NS_CHAIN_BLOCK=$cb timeout 300 python bench.py --trained-shape --steps 4 --warmup 2 --no-cpu-baseline --no-e2e --no-configs2 --extras-steps 3 2>$O/err.log | tail -1 > $O/b.json
  python - $tb $cb $O/b.json <<'P' | tee -a $O/ab.log
import json,sys
tb,cb,p=sys.argv[1:4]
try:
    d=json.load(open(p)); r=lambda x:round(x,2); s=d.get("serial",{})
    print("tail_bits",tb,"block",cb,"step",r(d["ms_per_step"]),"ms | in-step chain",r(d["kernel_ms"]["k_chain"]),"| chain alone",r((s.get("aligned_kernel_ms") or {}).get("k_chain",0)))
except Exception as ex:
    print(tb,cb,"FAILED",ex)
P
done; done
