#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03c
mkdir -p $OUT
cd $R
python -c "import __graft_entry__ as g; g.build()" || exit 1
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -q -x -k "split or multipass or deep_sweeps or c3_full or instantiation" > $OUT/tests.log 2>&1; echo "pytest rc $?"; tail -8 $OUT/tests.log
for S in 1 0; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --opt cd_split=$S > $OUT/bench_c3_split$S.json 2> $OUT/bench_c3_split$S.err || tail -3 $OUT/bench_c3_split$S.err
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
for S in 1 0; do
  sed "s/ds.set_option(\"profile\", 1)/ds.set_option(\"profile\", 1); ds.set_option(\"cd_split\", $S)/" tools/slab_c4_probe.py > /tmp/slab_probe_$S.py
  cp /tmp/slab_probe_$S.py tools/_slab_probe_tmp.py
  timeout -k 10 300 python tools/_slab_probe_tmp.py 8 2>&1 | tail -1 | sed "s/^/cd_split=$S: /"
done
rm -f tools/_slab_probe_tmp.py
timeout -k 10 300 bash tools/slab_trace.sh > $OUT/slab_trace.log 2>&1; tail -45 $OUT/slab_trace.log
echo STEP3_DONE
