#!/bin/bash
# VGPR / SGPR / scratch / LDS / occupancy of every kernel of the engine (compiler view, no GPU needed)
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -c -o /tmp/ns_res.o nanosim_amd/csrc/nanosim_amd.hip -Rpass-analysis=kernel-resource-usage "$@" 2>&1 |
  python3 -c "
import sys,re
cur=None; rows={}
for l in sys.stdin:
    m=re.search(r'Function Name: (\S+)',l)
    if m: cur=m.group(1); rows[cur]={}
    for k in ('VGPRs','AGPRs','SGPRs','ScratchSize \[bytes/lane\]','Occupancy \[waves/SIMD\]','LDS Size \[bytes/block\]'):
        m=re.search(k+r': (\d+)',l)
        if m and cur: rows[cur][k.split(' ')[0]]=m.group(1)
import subprocess
for k,v in rows.items():
    name=subprocess.run(['c++filt',k],capture_output=True,text=True).stdout.strip().split('(')[0]
    if name.startswith('void '): name=name[5:]
    if name.startswith('k_'): print('%-40s'%name, ' '.join('%s=%s'%(a,b) for a,b in v.items()))
"
