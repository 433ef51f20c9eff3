#!/bin/bash
# full -m gpu suite, then the default bench line and a kernel trace of it
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-check}
mkdir -p $OUT
cd $R
python -m pytest tests -m gpu -q --durations=5 > $OUT/tests.log 2>&1; tail -4 $OUT/tests.log
python bench.py --no-cpu-baseline > $OUT/bench_c3.json 2> $OUT/bench.err || exit 1
python -c "
import json
d=json.loads(open('$OUT/bench_c3.json').readline())
print('c3 value %.1f ms/step %.3f cd %.3f stats %.3f'%(d['value'],d['ms_per_step'],d['cd_kernel']['avg_launch_ms'],d['masked_gram']['avg_launch_ms']))"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 $R/bench.py --no-cpu-baseline > $OUT/prof.json 2> $OUT/prof.err
echo CHECK_DONE
