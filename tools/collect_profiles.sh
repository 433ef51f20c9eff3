#!/bin/bash
# copy what is to be judged from gpurun_out/$1 (tools/gpu_profile.sh) into profiles/$1 and regenerate profiles/traffic.json
TAG=${1:-r03}
R=$(dirname $(dirname $(readlink -f $0)))
S=$R/gpurun_out/$TAG; D=$R/profiles/$TAG
mkdir -p $D
# (gpurun merges new files INTO gpurun_out: remove the profile directories of earlier runs before calling tools/gpu_profile.sh,
# or the PMC averages mix runs; the kernel-trace tools take the newest file of a directory)
for f in bench_c1 bench_c2 bench_c3 bench_c3_grid bench_c3_s20w5 bench_c4 bench_c5 prof_c3 prof_c5; do cp $S/$f.json $D/ || exit 1; done
cp $(ls -t $S/prof_c3/*/*kernel_stats.csv | head -1) $D/c3_kernel_stats.csv
cp $(ls -t $S/prof_c5/*/*kernel_stats.csv | head -1) $D/c5_kernel_stats.csv
cp $S/c3_steady_iteration_timeline.txt $S/slab8_split0.log $S/slab8_split2.log $S/slab8_timeline_split0.txt $S/slab8_timeline_split2.txt $D/
python3 $R/tools/pmc_csv.py $S/pmc_fetch $S/pmc_write > $D/c3_pmc_summary.csv
python3 $R/tools/trace_avg.py $S/prof_c3 > $D/c3_trace_avg.txt 2>&1
cd $R && python3 tools/pmc_traffic.py c3 $S/pmc_fetch $S/pmc_write $(git rev-parse --short HEAD)
python3 - <<PY
import json
d = json.loads(open("$S/bench_c4.json").readline())
json.dump({"c4": {"n_gpus": 1, "value": d["value"], "unit": d["unit"], "steps": d["steps"], "warmup": d["warmup"],
                  "ms_per_step": d["ms_per_step"],
                  "source": "profiles/$TAG/bench_c4.json (python bench.py --workload c4 --steps 11 --warmup 1, one MI355X, this repository's run)"}},
          open("$R/profiles/single_gpu.json", "w"), indent=1)
PY
