#!/bin/bash
# round 6, GPU call 10: tile size of the SECOND record pass of -k (an event every ~300 bytes: byte-limited tiles) — 2 / 3 / 4 / 6 chunks per lane;
# parity of the new base first (DPP scan folded, run-window smear in k_hp_scan, per-mode tile template), then same-box A/B on configs[2]
cd "$(dirname "$0")/../.."
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r06j; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_metagenome.py -m gpu -x -q 2>&1 | tail -4 ) | tee $O/pytest_parity.log
cd /tmp; export TMPDIR=/tmp
for name in tf2 tf3 tf4 tf4w6 tf6w4 tf2; do
  f=$GRAFT_REPO_ROOT/nanosim_amd/_variants/$name.so
  rm -rf /tmp/prof_$name
  NANOSIM_AMD_LIB=$f timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -o s -- python $GRAFT_REPO_ROOT/bench.py --genome chr1 --fastq --kmer-bias 5 --aligned-only --steps 4 --warmup 1 --no-cpu-baseline --no-e2e --no-extras 2>$GRAFT_REPO_ROOT/$O/err_$name.log | tail -1 > $GRAFT_REPO_ROOT/$O/bench_$name.json
  python - $name $GRAFT_REPO_ROOT/$O/bench_$name.json /tmp/prof_$name <<'P' | tee -a $GRAFT_REPO_ROOT/$O/ab.log
import json,sys,glob,csv
name,p,prof=sys.argv[1:4]
r=lambda x:round(x,2)
ks={}
for f in glob.glob(prof+"/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        nm=row["Name"].split("(")[0].replace("void ","")
        if nm.startswith("k_hp_s") or "materialise<" in nm or "qualities<true" in nm: ks[nm]=r(float(row["AverageNs"])/1e6)
try:
    d=json.load(open(p)); print(name,"step",r(d["ms_per_step"]),"ms |",{k:r(v) for k,v in d["kernel_ms"].items() if v>0.01},"|",ks)
except Exception as ex:
    print(name,"FAILED",ex,ks)
P
done
