#!/bin/bash
# round 4, GPU call 3d: the CLI's device time per step (5 steps, outputs dropped behind PCIe) next to bench.py's errlog_on; configs[3] bench line
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04g; mkdir -p $O; cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
( NS_CLI_TRACE=1 NS_CLI_DROP_OUTPUT=1 timeout 300 python scripts/bench_cli.py -n 6000000 ) > $O/bench_cli_drop_output.log 2>&1; tail -14 $O/bench_cli_drop_output.log
timeout 300 python bench.py --no-cpu-baseline --no-e2e --no-configs2 > $O/bench_ecoli_extras.json 2> $O/bench_ecoli_extras.err
python - <<PY
import json
d = json.load(open("$O/bench_ecoli_extras.json"))
e = d["errlog_on"]; print("errlog_on", e["value"] / 1e6, e["ms_per_step"], "aligned", e["aligned_device_ms"], "unaligned", e["unaligned_device_ms"], "k_errlog", e["k_errlog_ms"], "frac", e["k_errlog_frac"])
PY
timeout 900 python bench.py --genome grch38 --chimeric --no-e2e --cpu-sample 2000 > $O/bench_grch38_chimeric.json 2> $O/bench_grch38.err; cut -c1-400 $O/bench_grch38_chimeric.json; tail -2 $O/bench_grch38.err
