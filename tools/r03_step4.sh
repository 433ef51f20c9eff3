#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03d
mkdir -p $OUT
cd $R
python -c "import __graft_entry__ as g; g.build()" || exit 1
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -q -x -k "split or multipass or deep_sweeps or instantiation" > $OUT/tests.log 2>&1; echo "pytest rc $?"; tail -5 $OUT/tests.log
for S in 0 2; do
  timeout -k 10 300 python bench.py --workload c2 --no-cpu-baseline --opt cd_split=$S > $OUT/bench_c2_split$S.json 2> $OUT/bench_c2_split$S.err || tail -3 $OUT/bench_c2_split$S.err
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/bench_*.json")):
    try:
        d=json.loads(open(f).readline()); c=d["cd_kernel"]; r=d["roofline"]
        print(f.split("/")[-1], "value %.1f ms/step %.3f cd %.3f stats %.3f steady cd %.3f stats %.3f"%(d["value"],d["ms_per_step"],c["avg_launch_ms"],d["masked_gram"]["avg_launch_ms"],r["steady_state"]["avg_launch_ms_parts"]["sweeps"],r["steady_state"]["avg_launch_ms_parts"]["statistics"]))
    except Exception as e:
        print(f, "unreadable", e)
PY
timeout -k 10 600 bash tools/ab_variants.sh "exechead:-DINSIDER_REG_EXEC_HEAD=1" 2>&1 | tail -8
echo STEP4_DONE
