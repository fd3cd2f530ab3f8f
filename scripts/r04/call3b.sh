#!/bin/bash
# round 4, GPU call 3b: k_errlog with LDS-staged, coalesced output — parity, then the errlog_on object of the bench line
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04e; mkdir -p $O; cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
( timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_transcriptome.py -m gpu -x -q 2>&1 | tail -8 ) > $O/pytest_parity.log 2>&1; cat $O/pytest_parity.log
timeout 300 python bench.py --no-cpu-baseline --no-e2e --no-configs2 > $O/bench_ecoli_extras.json 2> $O/bench_ecoli_extras.err
python - <<PY
import json
d = json.load(open("$O/bench_ecoli_extras.json"))
print("headline", round(d["value"] / 1e6, 1), "M reads/s", d["ms_per_step"])
print("serial", d["serial"]["value"] / 1e6, d["serial"]["ms_per_step"])
e = d["errlog_on"]; print("errlog_on", e["value"] / 1e6, e["ms_per_step"], "aligned", e["aligned_device_ms"], "k_errlog", e["k_errlog_ms"], "frac", e["k_errlog_frac"])
PY
