#!/bin/bash
# Closing job of a round (one gpurun call, AFTER profiles/traffic.json and profiles/issue.json were re-taken on the final library; tools/round_final.sh r05):
# the bench lines that carry the PMC figures, the whole GPU test suite, randomised parity sweeps.  Outputs under gpurun_out/<round>.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r05}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
python3 -c "import __graft_entry__ as g; g.build()" || exit 1
B="python3 $R/bench.py"
$B > $OUT/bench_c3_final.json 2> $OUT/final.err || exit 1
$B --steps 20 --warmup 5 > $OUT/bench_c3_s20w5_final.json 2>> $OUT/final.err || exit 1
$B --latent 40 --steps 11 --no-cpu-baseline > $OUT/bench_c3_K40_final.json 2>> $OUT/final.err || exit 1
$B --workload c1 --steps 121 --no-cpu-baseline > $OUT/bench_c1_s121.json 2>> $OUT/final.err || exit 1
$B --workload c5 --opt list_fine=0 --no-cpu-baseline > $OUT/bench_c5_list16.json 2>> $OUT/final.err || exit 1
echo "bench lines done"
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $OUT/gpu_tests.log 2>&1; echo "gpu tests rc $?"; tail -3 $OUT/gpu_tests.log
timeout -k 10 400 python3 tests/fuzz_parity.py ${FUZZ_N:-1500} ${FUZZ_SEED:-20261005} > $OUT/fuzz_final1500.txt 2>&1; echo "fuzz_parity rc $?"; tail -2 $OUT/fuzz_final1500.txt
timeout -k 10 200 python3 tests/fuzz_cd.py 2500 ${FUZZ_CD_SEED:-101} > $OUT/fuzz_cd_final.txt 2>&1; echo "fuzz_cd rc $?"; tail -2 $OUT/fuzz_cd_final.txt
echo FINAL_DONE
