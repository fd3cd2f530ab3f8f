#!/bin/bash
# round 6, GPU call 60: k_names composing through LDS stores (not flat stores): parity of all modes, then genome / transcriptome / metagenome benches
cd "$(dirname "$0")/../.."
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r06bp; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
( timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_metagenome.py tests/test_gpu_transcriptome.py tests/test_gpu_cli.py -m gpu -x -q 2>&1 | tail -3 ) | tee $O/pytest.log
for rep in 1 2 3; do
  timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-e2e --no-configs2 --no-extras 2>$O/err.log | tail -1 > $O/b.json
  python - "genome" $O/b.json <<'P' | tee -a $O/ab.log
import json,sys
name,p=sys.argv[1:3]
d=json.load(open(p)); r=lambda x:round(x,3); a=d["aligned_batch"]; k=a["kernel_ms"]
print("%-8s"%name,"step",r(d["ms_per_step"]),"ms",r(d["value"]/1e6),"M/s frac",r(d["roofline"]["frac"]),"| aligned call",r(a["device_ms"]),"plan",r(k["plan(k_nseg+k_lengths+scan+sort)"]),"chain",r(k["k_chain"]),"record stage",r(k["k_materialise"]),"kernel",r(d["roofline"]["kernel_ms"]))
P
done
for rep in 1 2; do
  echo "trx $(timeout 300 python scripts/bench_transcriptome.py 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["ms_per_batch"],3), "ms", round(d["value"]/1e6,1), "M reads/s", {k:round(v,3) for k,v in d["kernel_ms"].items() if v>0.01})')" | tee -a $O/ab.log
  timeout 300 python bench.py --metagenome --steps 6 --warmup 2 --no-cpu-baseline --no-e2e --no-extras 2>$O/err.log | tail -1 > $O/b.json
  python -c "
import json
d=json.load(open('$O/b.json')); r=lambda x:round(x,3)
print('zymo10', r(d['ms_per_step']), 'ms', r(d['value']/1e6), 'M reads/s', {k:r(v) for k,v in d['kernel_ms'].items() if v>0.01}, 'aligned call', r(d['aligned_batch']['device_ms']))" | tee -a $O/ab.log
done
