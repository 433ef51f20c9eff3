#!/bin/bash
# builds variants of the library on the GPU box and runs tools/k20_probe.py with each
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/k20
mkdir -p $OUT
cd $R
python -c "import __graft_entry__ as g; g.build()" || exit 1
SRC=insider_amd/csrc/insider_hip.hip
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Wno-pass-failed -mllvm -amdgpu-mfma-vgpr-form=1"
build() { # name, extra flags...
  n=$1; shift
  /opt/rocm/bin/hipcc $FLAGS "$@" -o $OUT/lib_$n.so $SRC -L/opt/rocm/lib -lrccl > $OUT/build_$n.log 2>&1 || { echo "build $n failed"; tail -5 $OUT/build_$n.log; }
}
build w3 -DINSIDER_REG_4WAVE_MAX=18 &
build nosgprvgpr -mllvm -amdgpu-spill-sgpr-to-vgpr=0 &
build nocap -DINSIDER_NO_CAP_COUNT=1 &
wait
for n in default w3 nosgprvgpr nocap; do
  echo "=== variant $n"
  if [ $n = default ]; then unset INSIDER_HIP_LIB; else export INSIDER_HIP_LIB=$OUT/lib_$n.so; fi
  timeout -k 10 200 python tools/k20_probe.py 2>&1 | grep -v "amdgpu.ids" | tee $OUT/probe_$n.log
done
echo K20_DONE
