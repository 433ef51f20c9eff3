#!/bin/bash
# experiment: multi-pass limits of the cold-iteration column solve (bench lines, c3): first limit, growth ratio, cold iterations
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/passes
mkdir -p $OUT
cd $R
for cfg in "64 4 3" "64 2 3" "64 3 3" "32 2 3" "128 2 3" "64 4 4" "64 2 5" "64 4 3"; do
  set -- $cfg
  python bench.py --no-cpu-baseline --opt cd_pass1=$1 --opt cd_pass_ratio=$2 --opt cd_cold_iters=$3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('pass1 $1 ratio $2 cold $3: value %.1f ms/step %.3f cd %.3f ms upd/s %.3g'%(d['value'],d['ms_per_step'],d['cd_kernel']['avg_launch_ms'],d['cd_kernel']['coordinate_updates_per_s']))" | tee -a $OUT/sweep.log
done
