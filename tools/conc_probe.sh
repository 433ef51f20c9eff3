#!/bin/bash
# Concurrent grid points on one GPU under different numbers of hardware queues (GPU_MAX_HW_QUEUES is read when the HIP
# runtime initialises): tools/conc_probe.sh <workload> <concurrent> <queues...>
R=${GRAFT_REPO_ROOT:-/root/repo}
W=$1; CK=$2; shift 2
mkdir -p $R/gpurun_out/r04
for Q in "$@"; do
  GPU_MAX_HW_QUEUES=$Q python3 $R/bench.py --workload $W --grid --concurrent $CK --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); g=d['grid']; c=d['grid_concurrent']
print('$W queues $Q: value %.1f it/s | grid serial %.2f s, k=$CK %.2f s, speedup %.2f, identical %s | per point serial %.0f ms conc %.0f ms'%(d['value'],g['wall_s'],c['wall_s'],c['speedup_vs_serial_grid'],c['identical_to_serial_grid'],g['per_point_ms']['optimize_call'],c['per_point_ms']['optimize_call']))"
done
