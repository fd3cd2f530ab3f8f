#!/bin/bash
# round 6, GPU call 4: k_hp_filter_w with its loads pipelined (NS_FILT_PF) / compiled for six waves, same-box A/B on configs[2]
cd "$(dirname "$0")/../.."
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r06d; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
export NS_HP_SCAN_KERNEL=1
for name in ${VARIANTS:-pf0 pf1 pf1w6 pf0w6 pf0 pf1}; do
  f=nanosim_amd/_variants/$name.so
  NANOSIM_AMD_LIB=$PWD/$f timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_$name -o x -- python bench.py --genome chr1 --fastq --kmer-bias 5 --aligned-only --steps 4 --warmup 1 --no-cpu-baseline --no-e2e --no-extras 2>$O/err_$name.log | tail -1 > $O/bench_$name.json
  python - $name $O/bench_$name.json $O/prof_$name <<'P' | tee -a $O/ab.log
import json,sys,glob,csv
name,p,prof=sys.argv[1:4]
try:
    d=json.load(open(p))
except Exception as ex:
    print(name,"FAILED",ex); sys.exit(0)
r=lambda x:round(x,2)
ks={}
for f in glob.glob(prof+"/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        nm=row["Name"].split("(")[0].replace("void ","")
        if nm.startswith("k_hp") or "materialise<" in nm or "qualities" in nm: ks[nm]=r(float(row["AverageNs"])/1e6)
print(name,"step",r(d["ms_per_step"]),"ms | aligned",r(d["aligned_batch"]["device_ms"]),{k:r(v) for k,v in d["kernel_ms"].items() if v>0.01},"|",ks)
P
done
