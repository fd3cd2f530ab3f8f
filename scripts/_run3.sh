R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02b; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp; ulimit -c 0
ARGS="--fastq --kmer-bias 5"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-genome-run $ARGS > $O/bench_stats.log 2>&1
find $O/stats -name "*kernel_stats.csv" | head -1 | xargs head -8 | cut -c1-100
PB="python $R/bench.py --steps 1 --warmup 1 --reads 200000 --no-cpu-baseline --no-genome-run $ARGS"
pmc() { timeout 300 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $O/pmc_$1 -o p -- $PB > $O/pmc_$1.log 2>&1; }
pmc sq1 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
pmc sq2 "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM"
python - $O <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].replace("void ", "").split("(")[0]
        if k.startswith("k_hp") or k.startswith("k_mat"): acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(k, " ".join("%s=%.1f" % (c.replace("SQ_", ""), sum(x) / len(x) / 200000) for c, x in sorted(v.items())))
PY
