#!/bin/bash
# experiment: multi-pass limits of the cold-iteration column solve (bench lines, c3)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/passes
mkdir -p $OUT
cd $R
for cfg in "64 4 3" "64 4 4" "64 3 4" "128 4 4" "256 4 4" "64 8 4" "48 4 4" "64 4 6"; do
  set -- $cfg
  python bench.py --no-cpu-baseline --opt cd_pass1=$1 --opt cd_pass_ratio=$2 --opt cd_cold_iters=$3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('pass1 $1 ratio $2 cold $3: value %.1f ms/step %.3f cd %.3f ms upd/s %.3g'%(d['value'],d['ms_per_step'],d['cd_kernel']['avg_launch_ms'],d['cd_kernel']['coordinate_updates_per_s']))" | tee -a $OUT/sweep.log
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 $R/bench.py --no-cpu-baseline --opt cd_pass1=64 --opt cd_pass_ratio=4 --opt cd_cold_iters=5 > $OUT/prof.json 2> $OUT/prof.err
