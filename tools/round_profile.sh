#!/bin/bash
# Profile job of a round (one gpurun call, AFTER the last source change; INSIDER_COMMIT=<hash> tools/round_profile.sh r05): rocprofv3 kernel statistics of the default command, the
# FETCH / WRITE passes behind profiles/traffic.json, the SQ issue-counter passes behind profiles/issue.json, the steady-state
# timeline, the c4 / 8 slab, concurrent grids.  Outputs under gpurun_out/$1 (default r04); tools/collect_round.sh copies them.
set -o pipefail
TAG=${1:-r05}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export GPU_MAX_HW_QUEUES=32
cd /tmp && export TMPDIR=/tmp
python3 -c "import sys; sys.path.insert(0, '$R'); import __graft_entry__ as g; g.build()" || exit 1
B="python3 $R/bench.py"
rm -rf $OUT/prof_c3 $OUT/prof_c5 $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_issue_c3_1 $OUT/pmc_issue_c3_2 $OUT/pmc_issue_c3_3 $OUT/pmc_issue_c3_K40_1 $OUT/pmc_issue_c3_K40_2 $OUT/pmc_issue_c3_K40_3 $OUT/pmc_issue_c5_1 $OUT/pmc_issue_c5_2 $OUT/pmc_issue_c5_3
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_c3 -- $B --no-cpu-baseline > $OUT/prof_c3.json 2> $OUT/prof_c3.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_c5 -- $B --workload c5 --no-cpu-baseline > $OUT/prof_c5.json 2> $OUT/prof_c5.err || exit 1
echo "kernel traces done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $B --steps 4 --warmup 0 --no-cpu-baseline > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.err || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $B --steps 4 --warmup 0 --no-cpu-baseline > $OUT/pmc_write.json 2> $OUT/pmc_write.err || exit 1
echo "traffic passes done"
bash $R/tools/pmc_issue.sh $TAG c3 > $OUT/pmc_issue.log 2>&1 || { tail -5 $OUT/pmc_issue.log; exit 1; }
bash $R/tools/pmc_issue.sh $TAG c3 "--latent 40" c3_K40 > $OUT/pmc_issue_K40.log 2>&1 || { tail -5 $OUT/pmc_issue_K40.log; exit 1; }
bash $R/tools/pmc_issue.sh $TAG c5 > $OUT/pmc_issue_c5.log 2>&1 || { tail -5 $OUT/pmc_issue_c5.log; exit 1; }
echo "issue passes done"
cd $R
python3 tools/iter_timeline.py $OUT/prof_c3 6 > $OUT/c3_steady_iteration_timeline.txt 2>&1
python3 tools/trace_avg.py $OUT/prof_c3 > $OUT/c3_trace_avg.txt 2>&1
python3 tools/slab_c4_probe.py 8 2>&1 | tail -1 > $OUT/slab8.log; cat $OUT/slab8.log
bash tools/slab_trace.sh 0 > $OUT/slab8_timeline.txt 2>&1; tail -1 $OUT/slab8_timeline.txt
bash tools/conc_probe.sh c1 6 32 2>&1 | tee $OUT/concurrent_grids.log
bash tools/conc_probe.sh c2 4 32 2>&1 | tee -a $OUT/concurrent_grids.log
bash tools/conc_probe.sh c3 2 32 2>&1 | tee -a $OUT/concurrent_grids.log
echo PROFILE_DONE
