#!/bin/bash
# Timing experiments on build variants of the sweep kernel (tools/exp_probe.sh "name:-DFLAG" ...): the fixed-sweep-count probe of
# tools/cd_probe.py (lone waves B = 2, 4; full machine B = 32768) per variant; variants named ok_* also run the c3 bench line.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/exp
mkdir -p $OUT
cd $R
python -c "import __graft_entry__ as g; g.build()" || exit 1
SRC=insider_amd/csrc/insider_hip.hip
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Wno-pass-failed -mllvm -amdgpu-mfma-vgpr-form=1"
names="default"
for spec in "$@"; do
  n=${spec%%:*}; f=${spec#*:}
  /opt/rocm/bin/hipcc $FLAGS $f -o $OUT/lib_$n.so $SRC -L/opt/rocm/lib -lrccl > $OUT/build_$n.log 2>&1 &
  names="$names $n"
done
wait
names="$names default"
for n in $names; do
  if [ $n = default ]; then unset INSIDER_HIP_LIB; else export INSIDER_HIP_LIB=$OUT/lib_$n.so; fi
  echo "== $n"
  python tools/cd_probe.py probe ${PROBE_K:-30} 2>&1 | grep "B="
  case $n in ok_*|default)
  for rep in 1 2; do
  python bench.py --no-cpu-baseline ${BENCH_ARGS} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']['steady_state']['avg_launch_ms_parts']
print('$n: value %.1f ms/step %.3f | cd %.3f ms stats %.4f | steady cd %.3f stats %.3f | G updates/s %.1f | loss %.9g'%(d['value'],d['ms_per_step'],d['cd_kernel']['avg_launch_ms'],d['masked_gram']['avg_launch_ms'],r['sweeps'],r['statistics'],d['cd_kernel']['coordinate_updates_per_s']/1e9,d['loss']))"
  done;;
  esac
done
echo EXP_DONE
