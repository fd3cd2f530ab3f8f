#!/bin/bash
# round 6, GPU call 51: k_names composing the header in LDS rows and copying it out in contiguous stores (names2) against the commit before (base):
# parity (the names are part of every record image the tests compare), then alternating bench lines
cd "$(dirname "$0")/../.."
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r06bf; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
( timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_metagenome.py tests/test_gpu_transcriptome.py tests/test_gpu_cli.py -m gpu -x -q 2>&1 | tail -3 ) | tee $O/pytest.log
for rep in 1 2 3 4; do for v in base names2; do
  NANOSIM_AMD_LIB=$PWD/nanosim_amd/_variants/$v.so timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-e2e --no-configs2 --no-extras 2>$O/err.log | tail -1 > $O/b.json
  python - "$v" $O/b.json <<'P' | tee -a $O/ab.log
import json,sys
name,p=sys.argv[1:3]
d=json.load(open(p)); r=lambda x:round(x,3); a=d["aligned_batch"]; k=a["kernel_ms"]
print("%-8s"%name,"step",r(d["ms_per_step"]),"ms",r(d["value"]/1e6),"M/s frac",r(d["roofline"]["frac"]),"| aligned call",r(a["device_ms"]),"chain",r(k["k_chain"]),"record stage",r(k["k_materialise"]),"kernel",r(d["roofline"]["kernel_ms"]))
P
done; done
