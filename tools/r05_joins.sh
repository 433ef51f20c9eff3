#!/bin/bash
# round 5: fewer stream joins on the main chain (option join_lean) and Qfull behind Qheld, against round 4's joins, one box:
# parity tests, bit identity of five fits (the joins change no arithmetic), bench lines, the steady iteration's timeline.
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
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sharded.py -x -q -k "optimize_one or optimize_31 or continuous or statistics_paths or two_ranks_on_one_gpu or rccl_allreduce_single" 2>&1 | tail -3
for P in 0 1; do
  INSIDER_HIP_OPTIONS="join_lean=$P" python tools/ab_identity.py run joins$P 2>>$OUT/err.log | tail -2
done
python tools/ab_identity.py cmp joins1 joins0
for r in 1 2; do
for P in 0 1 2; do
  python bench.py --no-cpu-baseline --opt join_lean=$P 2>>$OUT/err.log | tee $OUT/c3_j$P.json | line "c3 31 steps join_lean=$P"
  python bench.py --no-cpu-baseline --steps 20 --warmup 5 --opt join_lean=$P 2>>$OUT/err.log | tee $OUT/c3d_j$P.json | line "c3 s20w5 join_lean=$P"
  python bench.py --no-cpu-baseline --workload c2 --opt join_lean=$P 2>>$OUT/err.log | tee $OUT/c2_j$P.json | line "c2 join_lean=$P"
  python bench.py --no-cpu-baseline --workload c4 --steps 20 --warmup 5 --opt join_lean=$P 2>>$OUT/err.log | tee $OUT/c4_j$P.json | line "c4 s20w5 join_lean=$P"
done
done
bash tools/timeline.sh join1 --opt join_lean=1
bash tools/timeline.sh join0 --opt join_lean=0
cat gpurun_out/r05/timeline_join1.txt
echo JOINS_DONE
