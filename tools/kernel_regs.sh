#!/bin/bash
# VGPR / AGPR / SGPR / occupancy / LDS / scratch per kernel of one source file (hipcc -Rpass-analysis=kernel-resource-usage)
#   usage: tools/kernel_regs.sh cfun_amd/csrc/conv3d_wino.hip [name-filter]
SRC=$1; F=${2:-.}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Rpass-analysis=kernel-resource-usage -c $SRC -o /dev/null 2>&1 | python3 -c "
import sys,re
cur=None; rows=[]
for line in sys.stdin:
    m=re.search(r'Function Name: (\S+)',line)
    if m: cur={'name':m.group(1)}; rows.append(cur); continue
    for k,pat in (('vgpr',r' VGPRs: (\d+)'),('agpr',r'AGPRs: (\d+)'),('sgpr',r' SGPRs: (\d+)'),('occ',r'Occupancy \[waves/SIMD\]: (\d+)'),('scratch',r'ScratchSize \[bytes/lane\]: (\d+)'),('lds',r'LDS Size \[bytes/block\]: (\d+)')):
        m=re.search(pat,line)
        if m and cur is not None: cur[k]=m.group(1)
import subprocess
names=subprocess.run(['c++filt']+[r['name'] for r in rows],capture_output=True,text=True).stdout.split('\n')
for r,n in zip(rows,names):
    print('%-120s vgpr %3s agpr %3s sgpr %3s occ %s lds %6s scratch %s'%(n[:120],r.get('vgpr'),r.get('agpr'),r.get('sgpr'),r.get('occ'),r.get('lds'),r.get('scratch')))
" | grep -E "$F"
