#!/bin/bash
# A/B of build variants under the kernel-trace profiler: steady-state means of the row phase / statistics / sweeps / period
# (tools/iter_timeline.py) and the bench value, default and variant alternating on the same box.
#   tools/ab_prof.sh "name:-DFLAG ..." [bench.py args]
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/ab
mkdir -p $OUT
cd $R
python -c "import __graft_entry__ as g; g.build()" || exit 1
spec=$1; shift
n=${spec%%:*}; f=${spec#*:}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Wno-pass-failed -mllvm -amdgpu-mfma-vgpr-form=1"
/opt/rocm/bin/hipcc $FLAGS $f -o $OUT/lib_$n.so insider_amd/csrc/insider_hip.hip -L/opt/rocm/lib -lrccl > $OUT/build_$n.log 2>&1 || { tail $OUT/build_$n.log; exit 1; }
for v in default $n default $n default $n; do
  if [ $v = default ]; then unset INSIDER_HIP_LIB; else export INSIDER_HIP_LIB=$OUT/lib_$v.so; fi
  rm -rf $OUT/prof_$v
  (cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_$v -- python3 $R/bench.py --no-cpu-baseline "$@" > $OUT/prof_$v.json 2> $OUT/prof_$v.err)
  echo "$v: $(python3 -c "import json; print('value %.1f' % json.loads(open('$OUT/prof_$v.json').readline())['value'])") $(python3 tools/iter_timeline.py $OUT/prof_$v | tail -1 | cut -d: -f2-)"
done
