#!/bin/bash
# round 6, GPU call 28: counters of the wave-per-read unaligned chain (k_chain<true, true, 1>: 50 000 reads in 2.6 ms — where does it wait?)
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06ae; mkdir -p $O; cd /tmp; export TMPDIR=/tmp; ulimit -c 0
export HSA_ENABLE_IPC_MODE_LEGACY=0
export NS_UWIDE_SHIFT=31 K=2
B="python $R/scripts/r06/unaligned_probe.py"
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d /tmp/u1 -o p -- $B > /dev/null 2>&1
timeout 200 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM --kernel-trace --output-format csv -d /tmp/u2 -o p -- $B > /dev/null 2>&1
timeout 200 rocprofv3 --pmc SQ_INSTS_FLAT SQ_INSTS_GDS SQ_INSTS_EXP_GDS SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d /tmp/u3 -o p -- $B > /dev/null 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/u4 -o p -- $B > /dev/null 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/u5 -o p -- $B > /dev/null 2>&1
python3 - <<'P' | tee $O/ucoop_pmc.log
import csv,glob,collections
acc=collections.defaultdict(list); dur=[]
for d in ("u1","u2","u3","u4","u5"):
    for f in glob.glob('/tmp/%s/**/*counter_collection.csv'%d, recursive=True):
        for row in csv.DictReader(open(f)):
            if row['Kernel_Name'].startswith('void k_chain<true, true, 1>'): acc[row['Counter_Name']].append(float(row['Counter_Value']))
    for f in glob.glob('/tmp/%s/**/*kernel_trace.csv'%d, recursive=True):
        for row in csv.DictReader(open(f)):
            if row['Kernel_Name'].startswith('void k_chain<true, true, 1>'): dur.append((int(row['End_Timestamp'])-int(row['Start_Timestamp']))/1e6)
d={k:max(v) for k,v in acc.items()}
n=50000
print('ms', ' '.join('%.2f'%x for x in sorted(dur)[-6:]))
print({k: round(v/n,1) for k,v in d.items()})
print('per read: VALU %.0f SALU %.0f LDS %.0f VMEM_RD %.1f VMEM_WR %.1f | wait_any %.2f wait_inst %.2f active %.2f | lds conflict share %.2f | waves/SIMD %.2f' % (
    d['SQ_INSTS_VALU']/n, d['SQ_INSTS_SALU']/n, d['SQ_INSTS_LDS']/n, d['SQ_INSTS_VMEM_RD']/n, d['SQ_INSTS_VMEM_WR']/n,
    d['SQ_WAIT_ANY']/d['SQ_WAVE_CYCLES'], d['SQ_WAIT_INST_ANY']/d['SQ_WAVE_CYCLES'], d['SQ_ACTIVE_INST_ANY']/d['SQ_WAVE_CYCLES'],
    d['SQ_LDS_BANK_CONFLICT']/max(1,d['SQ_LDS_IDX_ACTIVE']), d['SQ_WAVE_CYCLES']*4/1024/(d['SQ_BUSY_CYCLES']/32)))
P
