#!/bin/bash
# VGPR / SGPR / scratch use of the kernels matching $1 (default: the register-resident sweep kernels) in the shipped library
R=$(dirname $(dirname $(readlink -f $0)))
T=$(mktemp -d); cp ${LIB:-$R/insider_amd/libinsider_hip.so} $T/lib.so; cd $T
/opt/rocm/lib/llvm/bin/llvm-objdump --offloading lib.so > /dev/null
/opt/rocm/lib/llvm/bin/llvm-readelf --notes lib.so.0.hipv4-amdgcn-amd-amdhsa--gfx950 | python3 -c "
import sys,re
pat=sys.argv[1]
cur={}
for line in sys.stdin:
    m=re.match(r'\s*-?\s*\.(\w+):\s*(.*)',line)
    if not m: continue
    k,v=m.group(1),m.group(2).strip()
    if k=='agpr_count' and cur.get('name'): pass
    cur[k]=v
    if k=='wavefront_size':
        if re.search(pat,cur.get('name','')): print(cur.get('name'), 'vgpr',cur.get('vgpr_count'),'sgpr',cur.get('sgpr_count'),'scratch',cur.get('private_segment_fixed_size'),'vgpr_spills',cur.get('vgpr_spill_count'),'sgpr_spills',cur.get('sgpr_spill_count'),'lds',cur.get('group_segment_fixed_size'))
        cur={}
" "${1:-k_cd_(cols|batch)_reg}"
rm -rf $T
