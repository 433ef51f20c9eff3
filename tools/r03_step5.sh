#!/bin/bash
# kernel trace of the default bench line (c3) with the timeline of one steady-state outer iteration, then the CPU model check
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03e
mkdir -p $OUT $R/gpurun_out/r03
cd $R
python -c "import __graft_entry__ as g; g.build()" || exit 1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_c3 -- python3 $R/bench.py --no-cpu-baseline > $OUT/prof_c3.json 2> $OUT/prof_c3.err || { tail -5 $OUT/prof_c3.err; }
cd $R
python tools/iter_timeline.py $OUT/prof_c3 6
python tools/trace_avg.py $OUT/prof_c3 31
python tools/cpu_validate.py gpurun_out/r03/cpu_model_check.json both 2>&1 | grep --line-buffered -v amdgpu.ids | cut -c1-700
echo STEP5_DONE
