#!/bin/bash
# quick kernel-trace profile of the default bench line.  usage (on the GPU box): tools/quick_prof.sh <tag> [bench.py args...]
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
OUT=$R/gpurun_out/${ROUND:-r04}/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/bench.py --no-cpu-baseline "$@" > $OUT/bench.json 2> $OUT/bench.err || exit 1
python3 $R/tools/kstats.py $OUT 24
