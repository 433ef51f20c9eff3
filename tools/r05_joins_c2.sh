#!/bin/bash
# round 5: which of the two join reductions costs c2 its 2.5 % (join_lean bits 1, 2), and c2's steady timeline with and without
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r05/joins
mkdir -p $OUT
cd $R
line() {
python -c "
import json,sys
d=json.loads(sys.stdin.readline()); mg=d['masked_gram']
print('$1: value %.1f ms/step %.3f | statistics %.3f ms | cd %.3f ms | rest %.3f ms | loss %.12g'%(d['value'],d['ms_per_step'],mg['avg_launch_ms'],d['cd_kernel']['avg_launch_ms'],d['ms_per_step']-mg['avg_launch_ms']-d['cd_kernel']['avg_launch_ms'],d['loss']))"
}
for r in 1 2; do
for P in 0 1 2 3; do
  python bench.py --no-cpu-baseline --workload c2 --opt join_lean=$P 2>>$OUT/err.log | tee $OUT/c2_j$P.json | line "c2 join_lean=$P"
done
done
bash tools/timeline.sh c2join3 --workload c2 --opt join_lean=3
bash tools/timeline.sh c2join0 --workload c2 --opt join_lean=0
cat gpurun_out/r05/timeline_c2join3.txt
cat gpurun_out/r05/timeline_c2join0.txt
