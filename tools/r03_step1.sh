#!/bin/bash
# round 3, first GPU call: full -m gpu suite on the new gene-order kernels / cap-hit counters, the bench lines that show
# cap hits and the longest solve per config, and the c4 / 8 slab timeline (baseline for the tail-hiding work)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03a
mkdir -p $OUT
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=8 > $OUT/tests.log 2>&1; echo "pytest rc $?"; tail -15 $OUT/tests.log
for W in c3 c2 c5; do
  timeout -k 10 300 python bench.py --workload $W --no-cpu-baseline > $OUT/bench_$W.json 2> $OUT/bench_$W.err || { echo "bench $W failed"; tail -5 $OUT/bench_$W.err; }
done
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_c3_s20w5.json 2> $OUT/bench_s20.err
timeout -k 10 300 python bench.py --workload c4 --steps 11 --warmup 1 --no-cpu-baseline > $OUT/bench_c4.json 2> $OUT/bench_c4.err
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/bench_*.json")):
    try:
        d=json.loads(open(f).readline())
        c=d["cd_kernel"]
        print(f.split("/")[-1], "value %.1f ms/step %.3f cd %.3f stats %.3f cap_hits %d max_gene_sweeps %d sweeps/gene/iter %.0f"%(d["value"],d["ms_per_step"],c["avg_launch_ms"],d["masked_gram"]["avg_launch_ms"],c["cap_hits"],c["max_gene_sweeps"],c["sweeps_per_gene_per_iter"]))
    except Exception as e:
        print(f, "unreadable", e)
PY
timeout -k 10 300 python tools/slab_c4_probe.py 8 > $OUT/slab8.log 2>&1; tail -2 $OUT/slab8.log
timeout -k 10 300 bash tools/slab_trace.sh > $OUT/slab_trace.log 2>&1; tail -70 $OUT/slab_trace.log
echo STEP1_DONE
