#!/bin/bash
# One gpurun call: the bench lines of every BASELINE config, the tune() grid (cold and warm-started), the rocprofv3 kernel
# summary of the default command and the two PMC passes behind profiles/traffic.json, the c4 / 8 slab probes.  Outputs under
# gpurun_out/$1 (default r03); copy what is to be judged into profiles/$1.
set -o pipefail
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 -c "import sys; sys.path.insert(0, '$R'); import __graft_entry__ as g; g.build()" || exit 1
python3 $R/bench.py > $OUT/bench_c3.json 2> $OUT/bench_c3.err || exit 1
python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_c3_s20w5.json 2>> $OUT/bench_c3.err || exit 1
python3 $R/bench.py --grid --no-cpu-baseline > $OUT/bench_c3_grid.json 2> $OUT/bench_grid.err || exit 1
for W in c1 c2 c5; do
  python3 $R/bench.py --workload $W --no-cpu-baseline > $OUT/bench_$W.json 2> $OUT/bench_$W.err || exit 1
done
python3 $R/bench.py --workload c4 --steps 11 --warmup 1 --no-cpu-baseline > $OUT/bench_c4.json 2> $OUT/bench_c4.err || exit 1
echo "benches done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_c3 -- python3 $R/bench.py --no-cpu-baseline > $OUT/prof_c3.json 2> $OUT/prof_c3.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_c5 -- python3 $R/bench.py --workload c5 --no-cpu-baseline > $OUT/prof_c5.json 2> $OUT/prof_c5.err || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --steps 4 --warmup 0 --no-cpu-baseline > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.err || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --steps 4 --warmup 0 --no-cpu-baseline > $OUT/pmc_write.json 2> $OUT/pmc_write.err || exit 1
echo "profiles done"
cd $R
python3 tools/iter_timeline.py $OUT/prof_c3 6 > $OUT/c3_steady_iteration_timeline.txt 2>&1
for S in 0 2; do
  python3 tools/slab_c4_probe.py --split $S 8 > $OUT/slab8_split$S.log 2>&1; tail -1 $OUT/slab8_split$S.log
done
bash tools/slab_trace.sh 2 > $OUT/slab8_timeline_split2.txt 2>&1; head -1 $OUT/slab8_timeline_split2.txt
bash tools/slab_trace.sh 0 > $OUT/slab8_timeline_split0.txt 2>&1; head -1 $OUT/slab8_timeline_split0.txt
echo PROFILE_DONE
