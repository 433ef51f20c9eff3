#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03f
mkdir -p $OUT
cd $R
python -c "import __graft_entry__ as g; g.build()" || exit 1
timeout -k 10 1100 python -m pytest tests -m gpu -q --durations=10 > $OUT/tests.log 2>&1; echo "pytest rc $?"; tail -22 $OUT/tests.log
for W in c3 c2 c5; do
  timeout -k 10 300 python bench.py --workload $W --no-cpu-baseline > $OUT/bench_$W.json 2> $OUT/bench_$W.err || { echo "bench $W failed"; tail -5 $OUT/bench_$W.err; }
done
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_c3_s20w5.json 2> $OUT/bench_s20.err
timeout -k 10 300 python bench.py --no-cpu-baseline --opt row_fused=0 > $OUT/bench_c3_nofuse.json 2> $OUT/bench_nofuse.err
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/bench_*.json")):
    try:
        d=json.loads(open(f).readline())
        c=d["cd_kernel"]; r=d["roofline"]["steady_state"]["avg_launch_ms_parts"]
        print(f.split("/")[-1], "value %.1f ms/step %.3f cd %.3f stats %.3f steady cd %.3f stats %.3f cap_hits %d max_gene_sweeps %d G/s %.1f"%(d["value"],d["ms_per_step"],c["avg_launch_ms"],d["masked_gram"]["avg_launch_ms"],r["sweeps"],r["statistics"],c["cap_hits"],c["max_gene_sweeps"],c["coordinate_updates_per_s"]/1e9))
    except Exception as e:
        print(f, "unreadable", e)
PY
echo STEP6_DONE
