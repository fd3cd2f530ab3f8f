#!/bin/bash
# round 6, GPU call 58: kernel timeline of a metagenome worker call (10^6 reads, aligned only)
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06bn; mkdir -p $O; cd /tmp; export TMPDIR=/tmp; ulimit -c 0
export HSA_ENABLE_IPC_MODE_LEGACY=0
rm -rf /tmp/tlm
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tlm -o p -- python $R/bench.py --metagenome --steps 3 --warmup 2 --no-cpu-baseline --no-e2e --no-extras --aligned-only > $O/bench.json 2>$O/err.log
python3 - <<'P' | tee $O/timeline.log
import csv,glob,re
rows=[]
for f in glob.glob('/tmp/tlm/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
starts=[i for i,r in enumerate(rows) if r[2].startswith('k_nseg')]
i0=starts[-1]
t0=rows[i0][0]
def short(n):
    n=re.sub(r'rocprim::ROCPRIM_\d+_NS::detail::','rp::',n); n=re.sub(r'\(.*','',n); n=n.replace('void ','')
    m=re.search(r'wrapped_(\w+?)_config',n)
    return ('rocprim:'+m.group(1)) if m else n[:60]
for s,e,n in rows[i0:]:
    if (e-s) > 15000 or n.startswith('k_'): print('%8.3f %8.3f  %s' % ((s-t0)/1e6,(e-s)/1e6,short(n)))
P
