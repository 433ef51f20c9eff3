#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03g
mkdir -p $OUT
cd $R
python -c "import __graft_entry__ as g; g.build()" || exit 1
timeout -k 10 1100 python -m pytest tests -m gpu -q --durations=6 > $OUT/tests.log 2>&1; echo "pytest rc $?"; tail -16 $OUT/tests.log
echo STEP7_DONE
