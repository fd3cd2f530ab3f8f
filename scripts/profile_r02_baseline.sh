#!/bin/bash
# round-2 starting point: bench lines, kernel-trace stats and PMC passes of the FASTQ / -hp -k 5 record path, phase ablations
# (NS_DEBUG_SKIP: 64 no quality LUT gathers, 128 no quality Philox, 256 hp_write without phase C, 512 without phase B,
# 1024 without hp_new_size).  Results under gpurun_out/r02a/.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02a; mkdir -p $O; cd /tmp; export TMPDIR=/tmp; ulimit -c 0
line() { grep -h "^{" $1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$2', round(d['value']/1e6,2), 'Mreads/s', round(d['ms_per_step'],2), 'ms', {k: round(v,2) for k,v in d['kernel_ms'].items() if v>0.01}, d.get('genome_run',{}).get('reads_per_s'))"; }
B="timeout 300 python $R/bench.py --no-cpu-baseline"
$B --steps 5 --warmup 2 > $O/bench_fasta.log 2>&1; line $O/bench_fasta.log fasta
$B --steps 3 --warmup 1 --no-genome-run --fastq > $O/bench_fastq.log 2>&1; line $O/bench_fastq.log fastq
$B --steps 3 --warmup 1 --no-genome-run --fastq --kmer-bias 5 > $O/bench_fastq_hp.log 2>&1; line $O/bench_fastq_hp.log fastq_hp
$B --steps 3 --warmup 1 --no-genome-run --kmer-bias 5 > $O/bench_hp.log 2>&1; line $O/bench_hp.log hp
for d in 64 128 192; do NS_DEBUG_SKIP=$d $B --steps 2 --warmup 1 --no-genome-run --fastq > $O/abl_fastq_$d.log 2>&1; line $O/abl_fastq_$d.log "fastq skip=$d"; done
for d in 256 512 1024 1792; do NS_DEBUG_SKIP=$d $B --steps 2 --warmup 1 --no-genome-run --fastq --kmer-bias 5 > $O/abl_hp_$d.log 2>&1; line $O/abl_hp_$d.log "fastq_hp skip=$d"; done
ARGS="--fastq --kmer-bias 5"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-genome-run $ARGS > $O/bench_stats.log 2>&1
find $O/stats -name "*kernel_stats.csv" | head -1 | xargs head -9 | cut -c1-110
PB="python $R/bench.py --steps 1 --warmup 1 --reads 200000 --no-cpu-baseline --no-genome-run $ARGS"
pmc() { timeout 300 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $O/pmc_$1 -o p -- $PB > $O/pmc_$1.log 2>&1; }
pmc sq1 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
pmc sq2 "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM"
pmc fetch "FETCH_SIZE"
pmc write "WRITE_SIZE"
python - $O <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].replace("void ", "").split("(")[0]
        if k.startswith("k_"): acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(k, " ".join("%s=%.1f" % (c.replace("SQ_", ""), sum(x) / len(x) / 200000) for c, x in sorted(v.items())))
PY
