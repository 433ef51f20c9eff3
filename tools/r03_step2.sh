#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03b
mkdir -p $OUT
cd $R
python -c "import __graft_entry__ as g; g.build()" || exit 1
timeout -k 10 200 python tools/k20_probe.py 2>&1 | grep -v amdgpu.ids | tee $OUT/k20.log
timeout -k 10 1000 python -m pytest tests -m gpu -q --durations=8 > $OUT/tests.log 2>&1; echo "pytest rc $?"; tail -25 $OUT/tests.log
for W in c3 c2; do
  timeout -k 10 300 python bench.py --workload $W --no-cpu-baseline > $OUT/bench_$W.json 2> $OUT/bench_$W.err || { echo "bench $W failed"; tail -5 $OUT/bench_$W.err; }
done
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_c3_s20w5.json 2> $OUT/bench_s20.err
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
echo STEP2_DONE
