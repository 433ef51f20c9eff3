#!/bin/bash
# round 5: the blocks of two coordinate steps (option cd_pairs) on latency-bound and throughput-bound workloads, one box:
# bit identity of five fits against cd_pairs = 0, the order-table / sweep parity tests, bench lines, the c4 / 8 slab
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r05/pairs
mkdir -p $OUT
cd $R
line() {
python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=((d['roofline'].get('steady_state') or {}).get('avg_launch_ms_parts') or {'sweeps':float('nan'),'statistics':float('nan')})
print('$1: value %.1f ms/step %.3f | cd %.3f ms | steady cd %.3f | G updates/s %.1f | loss %.12g'%(d['value'],d['ms_per_step'],d['cd_kernel']['avg_launch_ms'],r['sweeps'],d['cd_kernel']['coordinate_updates_per_s']/1e9,d['loss']))"
}
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "order_table or strong_cd or multipass or sweep_counts or instantiations" 2>&1 | tail -3
for P in 0 1; do
  export INSIDER_HIP_OPTIONS="cd_pairs=$P"
  python tools/ab_identity.py run pairs$P 2>>$OUT/err.log | tail -2
done
unset INSIDER_HIP_OPTIONS
python tools/ab_identity.py cmp pairs1 pairs0
for r in 1 2; do
for P in 0 1; do
  python bench.py --no-cpu-baseline --workload c1 --opt cd_pairs=$P 2>>$OUT/err.log | tee $OUT/c1_p$P.json | line "c1 pairs=$P"
  python bench.py --no-cpu-baseline --workload c1 --steps 121 --opt cd_pairs=$P 2>>$OUT/err.log | tee $OUT/c1_s121_p$P.json | line "c1 121 steps pairs=$P"
  python bench.py --no-cpu-baseline --workload c2 --opt cd_pairs=$P 2>>$OUT/err.log | tee $OUT/c2_p$P.json | line "c2 pairs=$P"
  python bench.py --no-cpu-baseline --steps 20 --warmup 5 --opt cd_pairs=$P 2>>$OUT/err.log | tee $OUT/c3_p$P.json | line "c3 s20w5 pairs=$P"
done
done
for P in 0 1; do
  INSIDER_HIP_OPTIONS="cd_pairs=$P" python tools/slab_c4_probe.py 8 2>>$OUT/err.log | sed "s/^/pairs=$P /"
done
echo PAIRS_DONE
