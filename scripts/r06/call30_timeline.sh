#!/bin/bash
# round 6, GPU call 30: kernel timeline of one bench step (both worker calls) from rocprofv3 --kernel-trace
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06ag; mkdir -p $O; cd /tmp; export TMPDIR=/tmp; ulimit -c 0
export HSA_ENABLE_IPC_MODE_LEGACY=0

rm -rf /tmp/tl
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o p -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-e2e --no-configs2 --no-extras > $O/bench.json 2>$O/err.log
python3 - <<'P' | tee $O/timeline.log
import csv,glob,re
rows=[]
for f in glob.glob('/tmp/tl/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id','?'), r.get('Stream_Id','?')))
rows.sort()
# the last step: from the last k_nseg launch with the big grid backwards... take the last 2 k_nseg (aligned + unaligned) as the start
starts=[i for i,r in enumerate(rows) if r[2].startswith('k_nseg')]
i0=starts[-2]
t0=rows[i0][0]
def short(n):
    n=re.sub(r'rocprim::ROCPRIM_\d+_NS::detail::','rp::',n); n=re.sub(r'\(.*','',n); n=n.replace('void ','')
    m=re.search(r'wrapped_(\w+?)_config',n)
    return ('rocprim:'+m.group(1)) if m else n[:60]
for s,e,n,q,st in rows[i0:]:
    print('%8.3f %8.3f  q%-3s s%-3s %s' % ((s-t0)/1e6,(e-s)/1e6,q,st,short(n)))
P
tail -2 $O/bench.json | cut -c1-300
