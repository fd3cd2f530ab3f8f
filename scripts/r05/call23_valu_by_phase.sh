#!/bin/bash
# round 5, GPU call 23: WHERE the record kernel's 3 004 vector instructions per read are: SQ_INSTS_VALU / SALU / LDS of a 200 000-read
# aligned launch under the NS_DEBUG_SKIP ablation bits of the kernel (1: no sub-run loads / final pass, 2: no letters, 16: no 16-byte
# stores, 32: no merge, 64: no letter-word Philox, 8: no head / tail)
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05z; mkdir -p $O; cd /tmp; export TMPDIR=/tmp; ulimit -c 0
export HSA_ENABLE_IPC_MODE_LEGACY=0
B="python $R/bench.py --steps 1 --warmup 1 --reads 200000 --no-cpu-baseline --no-e2e --no-configs2 --no-extras --aligned-only"
for sk in 0 1 2 8 16 32 64 99 107; do
  NS_DEBUG_SKIP=$sk timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --kernel-trace --output-format csv -d $O/skip_$sk -o p -- $B > /dev/null 2>&1
done
python3 - <<'P' | tee $O/valu_by_phase.log
import csv,glob,collections,os
O=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r05z'
for sk in (0,1,2,8,16,32,64,99,107):
    acc=collections.defaultdict(list)
    for f in glob.glob('%s/skip_%d/**/*counter_collection.csv'%(O,sk), recursive=True):
        for row in csv.DictReader(open(f)):
            if row['Kernel_Name'].startswith('void k_materialise<false, 0>'): acc[row['Counter_Name']].append(float(row['Counter_Value']))
    d={k:sum(v)/len(v)/200000 for k,v in acc.items()}
    print('skip=%-4d'%sk, ' '.join('%s %.0f'%(k.replace('SQ_INSTS_',''),v) for k,v in sorted(d.items())))
P
