#!/bin/bash
# round 5: the small dense products of the row phase on k_mm_rows2 / k_mm_reduce2 (option mm_fast) against round 4's kernels, one box:
# parity tests with the option at its default, then bench lines (31-step c3, driver command, c1, c2, c5) with mm_fast = 0 / 1 / 2.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r05/mm
mkdir -p $OUT
cd $R
line() {
python -c "
import json,sys
d=json.loads(sys.stdin.readline()); mg=d['masked_gram']; r=((d['roofline'].get('steady_state') or {}).get('avg_launch_ms_parts') or {'sweeps':float('nan'),'statistics':float('nan')})
print('$1: value %.1f ms/step %.3f | statistics %.3f ms | cd %.3f ms (steady %.3f) | rest %.3f ms | loss %.12g'%(d['value'],d['ms_per_step'],mg['avg_launch_ms'],d['cd_kernel']['avg_launch_ms'],r['sweeps'],d['ms_per_step']-mg['avg_launch_ms']-d['cd_kernel']['avg_launch_ms'],d['loss']))"
}
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "optimize_one or optimize_31 or row_operator or unmasked_row or level_gram or fused_level or continuous or statistics_paths" 2>&1 | tail -3
for r in 1 2; do
for P in 0 1 2; do
  python bench.py --no-cpu-baseline --opt mm_fast=$P 2>>$OUT/err.log | tee $OUT/c3_m$P.json | line "c3 31 steps mm_fast=$P"
  python bench.py --no-cpu-baseline --steps 20 --warmup 5 --opt mm_fast=$P 2>>$OUT/err.log | tee $OUT/c3d_m$P.json | line "c3 s20w5 mm_fast=$P"
  python bench.py --no-cpu-baseline --workload c1 --steps 121 --opt mm_fast=$P 2>>$OUT/err.log | tee $OUT/c1_m$P.json | line "c1 121 steps mm_fast=$P"
  python bench.py --no-cpu-baseline --workload c2 --opt mm_fast=$P 2>>$OUT/err.log | tee $OUT/c2_m$P.json | line "c2 mm_fast=$P"
  python bench.py --no-cpu-baseline --workload c5 --opt mm_fast=$P 2>>$OUT/err.log | tee $OUT/c5_m$P.json | line "c5 mm_fast=$P"
done
done
echo MM_DONE
