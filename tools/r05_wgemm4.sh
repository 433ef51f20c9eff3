#!/bin/bash
# round 5: the level Gram GEMM of the row phase on v_mfma_f64_4x4x4 (k_wgemm4, option row_gemm4) against k_wgemm, one box:
# parity tests, bench lines, steady timelines of c3.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r05/wgemm4
mkdir -p $OUT
cd $R
line() {
python -c "
import json,sys
d=json.loads(sys.stdin.readline()); mg=d['masked_gram']
print('$1: value %.1f ms/step %.3f | statistics %.3f ms | cd %.3f ms | rest %.3f ms | loss %.12g'%(d['value'],d['ms_per_step'],mg['avg_launch_ms'],d['cd_kernel']['avg_launch_ms'],d['ms_per_step']-mg['avg_launch_ms']-d['cd_kernel']['avg_launch_ms'],d['loss']))"
}
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q -k "level_gram or optimize_31 or c3_structure or c2_structure or statistics_paths" 2>&1 | tail -3
for r in 1 2; do
for P in 0 1; do
  python bench.py --no-cpu-baseline --opt row_gemm4=$P 2>>$OUT/err.log | tee $OUT/c3_g$P.json | line "c3 31 steps row_gemm4=$P"
  python bench.py --no-cpu-baseline --steps 20 --warmup 5 --opt row_gemm4=$P 2>>$OUT/err.log | tee $OUT/c3d_g$P.json | line "c3 s20w5 row_gemm4=$P"
  python bench.py --no-cpu-baseline --workload c2 --opt row_gemm4=$P 2>>$OUT/err.log | tee $OUT/c2_g$P.json | line "c2 row_gemm4=$P"
  python bench.py --no-cpu-baseline --workload c4 --steps 20 --warmup 5 --opt row_gemm4=$P 2>>$OUT/err.log | tee $OUT/c4_g$P.json | line "c4 s20w5 row_gemm4=$P"
done
done
bash tools/timeline.sh wg1 --opt row_gemm4=1
bash tools/timeline.sh wg0 --opt row_gemm4=0
cat gpurun_out/r05/timeline_wg1.txt
echo WGEMM4_DONE
