#!/bin/bash
# copy what is to be judged from gpurun_out/r04 (tools/r04_measure.sh, tools/r04_profile.sh) into profiles/r04 and regenerate
# profiles/traffic.json, profiles/issue.json, profiles/single_gpu.json
R=$(dirname $(dirname $(readlink -f $0)))
S=$R/gpurun_out/r04; D=$R/profiles/r04
mkdir -p $D
for f in bench_c1 bench_c2 bench_c3 bench_c3_s20w5 bench_c4 bench_c5 bench_c3_K32 bench_c3_K33 bench_c3_K36 bench_c3_K40 bench_c3_K44 bench_c3_K47 bench_c3_K48 bench_c3_K63 bench_c3_ctns2 grid_c1_k4 grid_c2_k4 grid_c3_k2 prof_c3 prof_c5; do
  [ -f $S/$f.json ] && cp $S/$f.json $D/
done
cp $(ls -t $S/prof_c3/*/*kernel_stats.csv | head -1) $D/c3_kernel_stats.csv
cp $(ls -t $S/prof_c5/*/*kernel_stats.csv | head -1) $D/c5_kernel_stats.csv
cp $S/c3_steady_iteration_timeline.txt $S/c3_trace_avg.txt $S/slab8.log $S/slab8_timeline.txt $S/concurrent_grids.log $D/ 2>/dev/null
cp $S/c3_pmc_issue.csv $D/c3_pmc_issue.csv
cp $S/c3_K40_pmc_issue.csv $D/c3_K40_pmc_issue.csv 2>/dev/null
cp $S/c5_pmc_issue.csv $D/c5_pmc_issue.csv 2>/dev/null
cp $S/issue.json $R/profiles/issue.json
python3 $R/tools/pmc_csv.py $S/pmc_fetch $S/pmc_write > $D/c3_pmc_summary.csv
cd $R && python3 tools/pmc_traffic.py c3 $S/pmc_fetch $S/pmc_write $(git rev-parse --short HEAD)
python3 - <<PY
import json
d = json.loads(open("$S/bench_c4.json").readline())
json.dump({"c4": {"n_gpus": 1, "value": d["value"], "unit": d["unit"], "steps": d["steps"], "warmup": d["warmup"],
                  "ms_per_step": d["ms_per_step"],
                  "source": "profiles/r04/bench_c4.json (python bench.py --workload c4 --steps 11 --warmup 1, one MI355X, this repository's run)"}},
          open("$R/profiles/single_gpu.json", "w"), indent=1)
PY
ls $D
