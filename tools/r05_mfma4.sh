#!/bin/bash
# round 5: pair-count column statistics with the second product on v_mfma_f64_4x4x4 (option col_mfma4, k_col_paircnt4) against
# k_col_paircnt on one box: parity tests of the statistics and of whole fits with the option on, then bench lines of both.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r05/mfma4
mkdir -p $OUT
cd $R
line() {
python -c "
import json,sys
d=json.loads(sys.stdin.readline()); mg=d['masked_gram']
print('$1: value %.1f ms/step %.3f | statistics %.3f ms frac %.3f | cd %.3f ms | loss %.12g'%(d['value'],d['ms_per_step'],mg['avg_launch_ms'],mg['frac'],d['cd_kernel']['avg_launch_ms'],d['loss']))"
}
INSIDER_HIP_OPTIONS="col_mfma4=${MFMA4_TEST:-1}" timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q -k "masked_gram or optimize_col or statistics_paths or c3_structure or c2_structure or continuous or optimize_31 or golden" 2>&1 | tail -3
for r in 1 2; do
for P in ${MFMA4_SET:-0 1}; do
  python bench.py --no-cpu-baseline --steps 20 --warmup 5 --opt col_mfma4=$P 2>>$OUT/err.log | tee $OUT/c3_m$P.json | line "c3 s20w5 mfma4=$P"
  python bench.py --no-cpu-baseline --workload c2 --opt col_mfma4=$P 2>>$OUT/err.log | tee $OUT/c2_m$P.json | line "c2 mfma4=$P"
  python bench.py --no-cpu-baseline --workload c3 --ctns 2 --opt col_mfma4=$P 2>>$OUT/err.log | tee $OUT/ctns_m$P.json | line "c3 ctns2 mfma4=$P"
done
done
echo MFMA4_DONE
