#!/bin/bash
# quick A/B: the default bench line, key timings only.  usage: tools/quick_bench.sh [bench.py args...]
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for i in 1 2; do
python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('value %.1f ms/step %.3f | stats %.4f ms (frac %.3f) cd %.3f ms | loss %.6f'%(d['value'],d['ms_per_step'],d['masked_gram']['avg_launch_ms'],d['masked_gram']['frac'],d['cd_kernel']['avg_launch_ms'],d['loss']))"
done
