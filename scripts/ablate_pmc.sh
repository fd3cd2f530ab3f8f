#!/bin/bash
# VALU / SALU / LDS instruction counts of k_materialise per ablated phase (NS_DEBUG_SKIP bits: 1 phase A, 2 payload pass,
# 4 reference staging, 8 head/tail, 16 the 16-byte stores).  Profiling aid; results under gpurun_out/ablate/.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ablate
mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for d in ${@:-0 1 2 4 8 16 3 7 31}; do
  NS_DEBUG_SKIP=$d rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d $O/d$d -o p -- python $R/bench.py --steps 1 --warmup 1 --reads 200000 --no-cpu-baseline > $O/d$d.log 2>&1
  python - $O/d$d $d <<'PY'
import csv, glob, sys, collections
d, tag = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "materialise" not in k: continue
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); n[(k, row["Counter_Name"])] += 1
for k, v in acc.items():
    print("skip=%s" % tag, " ".join("%s=%.0f" % (c.replace("SQ_", ""), x / n[(k, c)] / 200000) for c, x in sorted(v.items())))
PY
done
