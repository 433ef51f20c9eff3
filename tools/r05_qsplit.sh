#!/bin/bash
# round 5: Qfull / Qheld in two parts (option q_split: the first beside the last block of the row phase) against one piece, one box:
# parity tests, bit identity of five fits (NOT expected identical: the split sums in two runs), bench lines, the steady iteration's timeline.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r05/qsplit
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
  INSIDER_HIP_OPTIONS="q_split=$P" python tools/ab_identity.py run qsplit$P 2>>$OUT/err.log | tail -2
done
python tools/ab_identity.py cmp qsplit1 qsplit0
for r in 1 2; do
for P in 0 1; do
  python bench.py --no-cpu-baseline --opt q_split=$P 2>>$OUT/err.log | tee $OUT/c3_j$P.json | line "c3 31 steps q_split=$P"
  python bench.py --no-cpu-baseline --steps 20 --warmup 5 --opt q_split=$P 2>>$OUT/err.log | tee $OUT/c3d_j$P.json | line "c3 s20w5 q_split=$P"
  python bench.py --no-cpu-baseline --workload c2 --opt q_split=$P 2>>$OUT/err.log | tee $OUT/c2_j$P.json | line "c2 q_split=$P"
  python bench.py --no-cpu-baseline --workload c4 --steps 20 --warmup 5 --opt q_split=$P 2>>$OUT/err.log | tee $OUT/c4_j$P.json | line "c4 s20w5 q_split=$P"
done
done
bash tools/timeline.sh qs1 --opt q_split=1
bash tools/timeline.sh qs0 --opt q_split=0
cat gpurun_out/r05/timeline_qs1.txt
echo QSPLIT_DONE
