#!/bin/bash
# kernel timeline of one steady-state outer iteration of a bench.py command under rocprofv3 --kernel-trace:
#   tools/timeline.sh LABEL [bench.py arguments]   ->  gpurun_out/<tag>/timeline_LABEL.txt   (tag: $INSIDER_TAG, default r05)
set -o pipefail
LBL=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${INSIDER_TAG:-r05}
mkdir -p $OUT
export GPU_MAX_HW_QUEUES=32
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/tl_$LBL
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tl_$LBL -- python3 $R/bench.py --no-cpu-baseline "$@" > $OUT/tl_$LBL.json 2> $OUT/tl_$LBL.err || { tail -5 $OUT/tl_$LBL.err; exit 1; }
cd $R
python3 tools/iter_timeline.py $OUT/tl_$LBL 6 > $OUT/timeline_$LBL.txt 2>&1
rm -rf $OUT/tl_$LBL
tail -1 $OUT/timeline_$LBL.txt
