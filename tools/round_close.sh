#!/bin/bash
# The closing bench lines of a round alone (they carry the PMC figures of profiles/traffic.json / issue.json: run AFTER those were
# re-taken and committed) and the replay-based multi-GPU projection.   tools/round_close.sh r05
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
$B --workload c5 --opt list_fine=0 --no-cpu-baseline > $OUT/bench_c5_list16.json 2>> $OUT/final.err || exit 1
echo "bench lines done"
python3 tools/scale_replay.py --steps 20 --warmup 5 > $OUT/scale_replay.log 2>&1 || { tail -5 $OUT/scale_replay.log; exit 1; }
tail -1 $OUT/scale_replay.log
for i in 1 2 3; do $B --steps 20 --warmup 5 --no-cpu-baseline 2>> $OUT/final.err | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('driver command, run $i on this box: %.1f it/s'%d['value'])"; done | tee $OUT/driver_cmd_repeat.log
echo CLOSE_DONE
