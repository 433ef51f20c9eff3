#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/slab8_s${1:-2}_f${2:-0.03}
rm -rf $OUT/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cat > /tmp/slab_run.py <<PY
import sys, numpy as np
sys.path.insert(0, "$R")
from insider_amd import api, workloads
p = workloads.CONFIGS["c4"][1]
w = workloads.make("c4", gene_range=(0, p // 8))
ds = api.InsiderData(w.X, w.levels, w.M_train, w.M_test)
ds.set_option("cd_split", ${1:-2})
ds.set_option("cd_long_frac", ${2:-0.03})
A0, C0 = workloads.init_factors(w.n_levels, w.K, p, 7)
ds.optimize(A0, np.asfortranarray(C0[:, : p // 8]), w.K, w.lam, w.lam, w.alpha, max_iter=30, global_tol=-1, seed=1)
ds.close()
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 /tmp/slab_run.py > $OUT/run.log 2>&1
python3 $R/tools/iter_timeline.py $OUT/prof 4 | head -60
