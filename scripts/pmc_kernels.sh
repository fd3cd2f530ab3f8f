#!/bin/bash
# per-read PMC counters of the record-writing kernels (k_materialise*, k_hp_*), 200k-read launch.
# usage: pmc_kernels.sh "<counter list>" [NS_DEBUG_SKIP] [extra bench.py arguments, e.g. "--kmer-bias 5 --fastq"]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmck
mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rm -rf $O/run
NS_DEBUG_SKIP=${2:-0} rocprofv3 --pmc $1 --kernel-trace --output-format csv -d $O/run -o p -- python $R/bench.py --steps 1 --warmup 1 --reads 200000 --no-cpu-baseline $3 > $O/run.log 2>&1
python - $O/run <<'PY'
import csv, glob, sys, collections
d = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].replace("void ", "").split("(")[0]
        if not any(x in k for x in ("materialise", "payload", "chain", "k_hp", "k_words")): continue
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); n[(k, row["Counter_Name"])] += 1
for k, v in acc.items():
    print(k, " ".join("%s=%.1f" % (c.replace("SQ_", ""), x / n[(k, c)] / 200000) for c, x in sorted(v.items())))
PY
