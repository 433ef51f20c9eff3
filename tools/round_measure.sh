#!/bin/bash
# Measurement job of a round (one gpurun call; tools/round_measure.sh r05): the bench lines of every config, the grids with concurrent grid points, the K range,
# continuous covariates at size, the issue-counter and traffic passes.  Outputs under gpurun_out/$1 (default r04).
set -o pipefail
TAG=${1:-r05}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 -c "import sys; sys.path.insert(0, '$R'); import __graft_entry__ as g; g.build()" || exit 1
B="python3 $R/bench.py"
$B > $OUT/bench_c3.json 2> $OUT/bench_c3.err || exit 1
$B --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_c3_s20w5.json 2>> $OUT/bench_c3.err || exit 1
echo "c3 done"
for W in c1 c2 c5; do
  $B --workload $W --no-cpu-baseline > $OUT/bench_$W.json 2> $OUT/bench_$W.err || exit 1
done
$B --workload c4 --steps 11 --warmup 1 --no-cpu-baseline > $OUT/bench_c4.json 2> $OUT/bench_c4.err || exit 1
$B --workload c4 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_c4_s20w5.json 2>> $OUT/bench_c4.err || exit 1
$B --workload c1 --steps 121 --no-cpu-baseline > $OUT/bench_c1_s121.json 2>> $OUT/bench_c1.err || exit 1
echo "configs done"
# tune() grids: serial, 2 and 4 grid points at a time (handles of one resident data set)
for W in c1 c2; do
  for CK in 2 4; do
    $B --workload $W --grid --concurrent $CK --no-cpu-baseline > $OUT/grid_${W}_k$CK.json 2> $OUT/grid_${W}_k$CK.err || exit 1
  done
done
$B --grid --concurrent 2 --no-cpu-baseline > $OUT/grid_c3_k2.json 2> $OUT/grid_c3_k2.err || exit 1
echo "grids done"
# K range at c3's shape; continuous covariates at c3's size
for KK in 32 33 36 40 44 47 48 63; do
  $B --latent $KK --steps 11 --no-cpu-baseline > $OUT/bench_c3_K$KK.json 2> $OUT/bench_c3_K$KK.err || exit 1
done
$B --ctns 2 --no-cpu-baseline > $OUT/bench_c3_ctns2.json 2> $OUT/bench_c3_ctns2.err || exit 1
echo "K range and ctns done"
echo MEASURE_DONE
