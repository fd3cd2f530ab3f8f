#!/bin/bash
# round 6, validation call: smoke(), the whole -m gpu suite, the driver's bench command, the configs[3] / configs[4] lines
cd "$(dirname "$0")/../.."
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/${TAG:-r06v}; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
( timeout 120 python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -2; timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 ) | tee $O/pytest_gpu.log
timeout 600 python bench.py --steps 10 --warmup 3 2>$O/bench_default.err | tail -1 > $O/bench_ecoli_fasta.json
python - $O/bench_ecoli_fasta.json <<'P' | tee $O/bench_summary.log
import json,sys
d=json.load(open(sys.argv[1])); r=lambda x:round(x,3)
print("step", r(d["ms_per_step"]), "ms", r(d["value"]/1e6), "M reads/s | aligned", r(d["aligned_batch"]["device_ms"]), {k:r(v) for k,v in d["kernel_ms"].items() if v>0.01}, "| unaligned", r(d["unaligned_batch"]["device_ms"]), {k:r(v) for k,v in d["unaligned_batch"]["kernel_ms"].items() if v>0.01})
print("roofline", {k:(r(v) if isinstance(v,float) else v) for k,v in d["roofline"].items() if k in ("frac","frac_kernel_only_bytes","frac_counter_bytes","whole_aligned_batch_frac","traffic_source")})
for k in ("serial","errlog_on"): print(k, {a:(r(b) if isinstance(b,float) else b) for a,b in d[k].items() if not isinstance(b,(dict,str))})
for key in ("configs2","chr1_fasta","trained_shape"):
    c=d.get(key,{}); print(key, r(c.get("ms_per_step",0)), r(c.get("value",0)/1e6), "M reads/s", {k:r(v) for k,v in (c.get("aligned_batch",{}).get("kernel_ms") or {}).items() if v>0.01}, "frac", c.get("roofline",{}).get("frac"), "batch frac", c.get("roofline",{}).get("whole_aligned_batch_frac"), c.get("roofline",{}).get("traffic_source"))
print("cpu", d.get("cpu_baseline",{}).get("value"), d.get("cpu_baseline",{}).get("cores"), "e2e", {k:(r(v.get("reads_per_s",0)/1e6) if isinstance(v,dict) and "reads_per_s" in v else None) for k,v in d.get("e2e",{}).items()})
P
timeout 400 python bench.py --genome grch38 --chimeric --steps 5 --warmup 2 --no-cpu-baseline --no-e2e 2>$O/bench_grch38.err | tail -1 > $O/bench_grch38_chimeric.json
timeout 300 python bench.py --metagenome --steps 5 --warmup 2 --no-cpu-baseline --no-e2e 2>$O/bench_meta.err | tail -1 > $O/bench_zymo10_metagenome.json
for f in bench_grch38_chimeric bench_zymo10_metagenome; do python -c "
import json,sys
d=json.load(open('$O/$f.json')); r=lambda x:round(x,3)
print('$f', r(d['ms_per_step']), 'ms', r(d['value']/1e6), 'M reads/s', {k:r(v) for k,v in d['kernel_ms'].items() if v>0.01}, 'serial', r(d.get('serial',{}).get('ms_per_step',0)), 'errlog_on', r(d.get('errlog_on',{}).get('ms_per_step',0)), r(d.get('errlog_on',{}).get('k_errlog_ms',0)), 'frac', r(d['roofline']['frac']))
" | tee -a $O/bench_summary.log; done
( timeout 300 python scripts/bench_transcriptome.py 2>&1 | tail -3; timeout 300 python scripts/bench_transcriptome.py --model-ir 2>&1 | tail -3 ) | tee $O/bench_transcriptome.log
