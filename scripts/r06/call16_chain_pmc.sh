#!/bin/bash
# round 6, GPU call 16: counters of k_chain<true, false> on the trained-shape model (hot-prefix image, 640-thread workgroups) against the default model
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06p; mkdir -p $O; cd /tmp; export TMPDIR=/tmp; ulimit -c 0
export HSA_ENABLE_IPC_MODE_LEGACY=0
for tag in default trained; do
  X=""; E=""
  [ $tag = trained ] && X="--trained-shape"
  [ $tag = trained256 ] && X="--trained-shape" && export NS_CHAIN_BLOCK=256
  [ $tag = trained384 ] && X="--trained-shape" && export NS_CHAIN_BLOCK=384
  B="python $R/bench.py $X --steps 1 --warmup 1 --reads 1000000 --no-cpu-baseline --no-e2e --no-configs2 --no-extras --aligned-only"
  timeout 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d /tmp/c1_$tag -o p -- $B > /dev/null 2>&1
  timeout 200 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM --kernel-trace --output-format csv -d /tmp/c2_$tag -o p -- $B > /dev/null 2>&1
  unset NS_CHAIN_BLOCK
done
python3 - <<'P' | tee $O/chain_pmc.log
import csv,glob,collections
for tag in ("default","trained"):
    acc=collections.defaultdict(list); dur=[]
    for d in ("c1","c2"):
        for f in glob.glob('/tmp/%s_%s/**/*counter_collection.csv'%(d,tag), recursive=True):
            for row in csv.DictReader(open(f)):
                if row['Kernel_Name'].startswith('void k_chain<true, false>'): acc[row['Counter_Name']].append(float(row['Counter_Value']))
        for f in glob.glob('/tmp/%s_%s/**/*kernel_trace.csv'%(d,tag), recursive=True):
            for row in csv.DictReader(open(f)):
                if row['Kernel_Name'].startswith('void k_chain<true, false>'): dur.append((int(row['End_Timestamp'])-int(row['Start_Timestamp']))/1e6)
    d={k:max(v) for k,v in acc.items()}      # the main launch (the retry passes are small)
    n=950000
    print(tag, 'ms', ' '.join('%.2f'%x for x in sorted(dur)[-4:]), '| per read: VALU %.0f SALU %.0f LDS %.0f VMEM_RD %.1f VMEM_WR %.1f | wait_any %.2f wait_inst %.2f active %.2f | lds conflict share %.2f | waves/SIMD %.2f' % (
        d['SQ_INSTS_VALU']/n, d['SQ_INSTS_SALU']/n, d['SQ_INSTS_LDS']/n, d['SQ_INSTS_VMEM_RD']/n, d['SQ_INSTS_VMEM_WR']/n,
        d['SQ_WAIT_ANY']/d['SQ_WAVE_CYCLES'], d['SQ_WAIT_INST_ANY']/d['SQ_WAVE_CYCLES'], d['SQ_ACTIVE_INST_ANY']/d['SQ_WAVE_CYCLES'],
        d['SQ_LDS_BANK_CONFLICT']/max(1,d['SQ_LDS_IDX_ACTIVE']), d['SQ_WAVE_CYCLES']*4/1024/(d['SQ_BUSY_CYCLES']/32)))
P
