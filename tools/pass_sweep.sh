#!/bin/bash
# Sweep of the multi-pass options of the cold outer iterations on the driver's command (python bench.py --steps 20 --warmup 5):
# cd_cold_iters x cd_pass1 x cd_pass_ratio, two runs each.   bash tools/pass_sweep.sh > gpurun_out/pass_sweep.log
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python3 -c "import __graft_entry__ as g; g.build()" || exit 1
run() {
  for rep in 1 2; do
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('%-60s value %.1f ms/step %.3f cd %.3f ms loss %.10g'%('$*',d['value'],d['ms_per_step'],d['cd_kernel']['avg_launch_ms'],d['loss']))"
  done
}
run
for CI in 3 4 5 6; do run --opt cd_cold_iters=$CI; done
for P1 in 32 48 96 128 256; do run --opt cd_pass1=$P1; done
for PR in 2 3 6 8; do run --opt cd_pass_ratio=$PR; done
run --opt cd_pass1=128 --opt cd_pass_ratio=3
run --opt cd_pass1=32 --opt cd_pass_ratio=3
run --opt cd_cold_iters=5 --opt cd_pass1=128
run
echo SWEEP_DONE
