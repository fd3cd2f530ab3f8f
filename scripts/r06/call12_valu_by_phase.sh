#!/bin/bash
# round 6, GPU call 12: WHERE the v7 record kernel's vector instructions are: SQ_INSTS_VALU / SALU / LDS per read of a 200 000-read aligned launch of
# the -DNS_ABLATE build under its NS_DEBUG_SKIP bits (1: no chunk loads / final step, 2: no letters, 8: no head / tail, 16: no stores, 32: no event sub-run
# merge, 64: no letter-word Philox, 128: no event sub-run loads, 256: no chunk-lane look-ups and loads, 512: chunk step without mask / OR)
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06l; mkdir -p $O; cd /tmp; export TMPDIR=/tmp; ulimit -c 0
export HSA_ENABLE_IPC_MODE_LEGACY=0
export NANOSIM_AMD_LIB=$R/nanosim_amd/_variants/ablate.so
B="python $R/bench.py --steps 1 --warmup 1 --reads 200000 --no-cpu-baseline --no-e2e --no-configs2 --no-extras --aligned-only"
SK="0 1 2 8 16 32 64 128 256 512 419 1023"
for sk in $SK; do
  NS_DEBUG_SKIP=$sk timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --kernel-trace --output-format csv -d /tmp/skip_$sk -o p -- $B > /dev/null 2>&1
done
SK="$SK" python3 - <<'P' | tee $O/valu_by_phase.log
import csv,glob,collections,os
for sk in [int(x) for x in os.environ["SK"].split()]:
    acc=collections.defaultdict(list); dur=[]
    for f in glob.glob('/tmp/skip_%d/**/*counter_collection.csv'%sk, recursive=True):
        for row in csv.DictReader(open(f)):
            if row['Kernel_Name'].startswith('void k_materialise<false, 0>'): acc[row['Counter_Name']].append(float(row['Counter_Value']))
    for f in glob.glob('/tmp/skip_%d/**/*kernel_trace.csv'%sk, recursive=True):
        for row in csv.DictReader(open(f)):
            if row['Kernel_Name'].startswith('void k_materialise<false, 0>'): dur.append((int(row['End_Timestamp'])-int(row['Start_Timestamp']))/1e6)
    d={k:sum(v)/len(v)/200000 for k,v in acc.items()}
    print('skip=%-5d'%sk, ' '.join('%s %.0f'%(k.replace('SQ_INSTS_',''),v) for k,v in sorted(d.items())), 'ms', ' '.join('%.3f'%x for x in dur))
P
