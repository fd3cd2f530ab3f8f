#!/bin/bash
# round 6, GPU call 77 (call 42 on the last build of the round): the large GPU == oracle sweeps on the last build (the wave-per-read unaligned chain and the metagenome pass order changed
# this round), then the default bench line with the record kernel bracketed by its own events
cd "$(dirname "$0")/../.."
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r06sw2; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
( timeout 1500 python scripts/parity_sweep.py 40000 2>&1 | tail -20 ) | tee $O/parity_sweep.log
( timeout 900 python scripts/parity_meta_big.py 20000 2>&1 | tail -8 ) | tee $O/parity_meta_big.log
( timeout 900 python scripts/parity_trx_big.py 2>&1 | tail -6 ) | tee $O/parity_trx_big.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-e2e 2>$O/err.log | tail -1 > $O/bench.json
python - $O/bench.json <<'P' | tee $O/bench_summary.log
import json,sys
d=json.load(open(sys.argv[1])); r=lambda x:round(x,3)
print("step", r(d["ms_per_step"]), "ms", r(d["value"]/1e6), "M reads/s", {k:r(v) for k,v in d["kernel_ms"].items() if v>0.01})
print("roofline", {k:(r(v) if isinstance(v,float) else v) for k,v in d["roofline"].items() if k in ("frac","frac_stage","kernel_ms","stage_ms","frac_kernel_only_bytes","frac_counter_bytes","achieved")})
for key in ("configs2","chr1_fasta","trained_shape"):
    c=d.get(key,{}); print(key, r(c.get("ms_per_step",0)), r(c.get("value",0)/1e6), {k:(r(v) if isinstance(v,float) else v) for k,v in c.get("roofline",{}).items() if k in ("frac","frac_stage","kernel_ms","stage_ms","kernel")})
P
