#!/bin/bash
# rocprofv3 kernel stats (+ one SQ counter pass) of one bench command: scripts/quick_stats.sh <tag> [bench.py arguments]; run on the GPU box (gpurun)
TAG=$1; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG
mkdir -p $O; cd /tmp; export TMPDIR=/tmp; ulimit -c 0
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-configs2 --no-genome-run "$@" > $O/bench.log 2>&1
head -12 $O/stats/*/s_kernel_stats.csv 2>/dev/null | cut -c1-150 || find $O -name "*kernel_stats.csv" | head -1 | xargs head -12 | cut -c1-150
if [ -n "$PMC" ]; then
  B="python $R/bench.py --steps 1 --warmup 1 --reads 200000 --no-cpu-baseline --no-e2e --no-configs2 --aligned-only $@"
  timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/pmc/sq1 -o p -- $B > /dev/null 2>&1
  python - <<PY
import csv, glob, collections
f = glob.glob("$O/pmc/sq1/**/p_counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for row in csv.DictReader(open(f[0])):
    acc[row["Kernel_Name"][:60]][row["Counter_Name"]] += float(row["Counter_Value"])
for k, v in acc.items():
    if v.get("SQ_INSTS_VALU", 0) > 1e6: print(k, {a: round(b / 200000, 1) for a, b in v.items()})
PY
fi
