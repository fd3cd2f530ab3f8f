#!/bin/bash
# round 6, GPU call 66 (call 61 on the two-key order build): kernel timeline of a configs[3] step (GRCh38-size reference, --chimeric)
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06bv; mkdir -p $O; cd /tmp; export TMPDIR=/tmp; ulimit -c 0
export HSA_ENABLE_IPC_MODE_LEGACY=0
rm -rf /tmp/tlg
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tlg -o p -- python $R/bench.py --genome grch38 --chimeric --steps 3 --warmup 2 --no-cpu-baseline --no-e2e --no-extras > $O/bench.json 2>$O/err.log
python3 - <<'P' | tee $O/timeline.log
import csv,glob,re
rows=[]
for f in glob.glob('/tmp/tlg/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id','?')))
rows.sort()
starts=[i for i,r in enumerate(rows) if r[2].startswith('k_nseg')]
i0=starts[-2]
t0=rows[i0][0]
def short(n):
    n=re.sub(r'rocprim::ROCPRIM_\d+_NS::detail::','rp::',n); n=re.sub(r'\(.*','',n); n=n.replace('void ','')
    m=re.search(r'wrapped_(\w+?)_config',n)
    return ('rocprim:'+m.group(1)) if m else n[:60]
for s,e,n,q in rows[i0:]:
    if (e-s) > 20000 or n.startswith('k_'): print('%8.3f %8.3f  q%-2s %s' % ((s-t0)/1e6,(e-s)/1e6,q,short(n)))
P
