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
i=0
for v in default $n default $n default $n; do
  i=$((i + 1))
  if [ $v = default ]; then unset INSIDER_HIP_LIB; else export INSIDER_HIP_LIB=$OUT/lib_$v.so; fi
  D=$OUT/prof_${v}_$i
  rm -rf $D
  (cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --output-format csv -d $D -- python3 $R/bench.py --no-cpu-baseline "$@" > $D.json 2> $D.err)
  python3 tools/iter_timeline.py $D > $D.timeline.txt 2>&1
  echo "$v ($i): $(python3 -c "import json; print('value %.1f' % json.loads(open('$D.json').readline())['value'])") $(tail -1 $D.timeline.txt | cut -d: -f2-)"
  rm -rf $D   # (the traces are 10 MB each: the timeline text is what is kept)
done
