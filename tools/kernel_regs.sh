#!/bin/bash
# registers / scratch of the kernels of the built library whose (mangled) name matches $1; $2 = directory for the extracted code
# object and its disassembly (default: a fresh one under /tmp)
PAT=${1:-k_}
D=${2:-$(mktemp -d /tmp/co.XXXX)}
R=$(cd "$(dirname "$0")/.." && pwd)
cp $R/insider_amd/libinsider_hip.so $D/ && cd $D || exit 1
/opt/rocm/lib/llvm/bin/llvm-objdump --offloading libinsider_hip.so > /dev/null
CO=$(ls | grep gfx950 | head -1)
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $CO > notes.txt
/opt/rocm/lib/llvm/bin/llvm-objdump -d $CO > dis.txt
python3 - "$PAT" <<'PY'
import sys, re
t = open('notes.txt').read()
for blk in t.split('- .agpr_count')[1:]:
    name = re.search(r'\.name:\s+(\S+)', blk).group(1)
    if sys.argv[1] in name:
        g = lambda k: re.search(r'\.%s:\s+(\d+)' % k, blk).group(1)
        print(name, 'vgpr', g('vgpr_count'), 'sgpr', g('sgpr_count'), 'spill', g('vgpr_spill_count'), 'scratch', g('private_segment_fixed_size'))
PY
echo $D
