#!/bin/bash
# A/B on one box: the library in the tree against tools/_ab/lib_old.so (the build of the commit before): bench lines of c3 (default
# and the driver's command), c2 and c1; fits dumped by tools/ab_identity.py compared bit for bit; the sweep-kernel parity tests.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/ab
mkdir -p $OUT
cd $R
line() {
python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=((d['roofline'].get('steady_state') or {}).get('avg_launch_ms_parts') or {'sweeps':float('nan'),'statistics':float('nan')})
print('$1 $2: value %.1f ms/step %.3f | cd %.3f ms stats %.4f | steady cd %.3f stats %.3f | G updates/s %.1f | loss %.12g | sha %s'%(d['value'],d['ms_per_step'],d['cd_kernel']['avg_launch_ms'],d['masked_gram']['avg_launch_ms'],r['sweeps'],r['statistics'],d['cd_kernel']['coordinate_updates_per_s']/1e9,d['loss'],d.get('library_source_sha')))"
}
for n in new old new old; do
  if [ $n = new ]; then unset INSIDER_HIP_LIB; else export INSIDER_HIP_LIB=$R/tools/_ab/lib_old.so; fi
  python bench.py --no-cpu-baseline 2>$OUT/err_$n.log | tee $OUT/bench_c3_$n.json | line $n c3
  python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>>$OUT/err_$n.log | tee $OUT/bench_c3_s20_$n.json | line $n c3_s20w5
done
for n in new old; do
  if [ $n = new ]; then unset INSIDER_HIP_LIB; else export INSIDER_HIP_LIB=$R/tools/_ab/lib_old.so; fi
  python bench.py --no-cpu-baseline --workload c2 2>>$OUT/err_$n.log | tee $OUT/bench_c2_$n.json | line $n c2
  python bench.py --no-cpu-baseline --workload c1 2>>$OUT/err_$n.log | tee $OUT/bench_c1_$n.json | line $n c1
  python bench.py --no-cpu-baseline --workload c5 2>>$OUT/err_$n.log | tee $OUT/bench_c5_$n.json | line $n c5
  python tools/ab_identity.py run $n 2>>$OUT/err_$n.log
done
unset INSIDER_HIP_LIB
python tools/ab_identity.py cmp new old
echo "identity rc $?"
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "strong_cd or multipass or sweep_counts or optimize_31 or instantiations or golden" 2>&1 | tail -5
echo AB_DONE
