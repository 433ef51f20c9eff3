#!/bin/bash
# copy what is to be judged from gpurun_out/<round> (tools/round_measure.sh, tools/round_profile.sh) into profiles/<round> and
# regenerate profiles/traffic.json, profiles/issue.json, profiles/single_gpu.json        usage: tools/collect_round.sh r05
TAG=${1:-r05}
R=$(dirname $(dirname $(readlink -f $0)))
S=$R/gpurun_out/$TAG; D=$R/profiles/$TAG
mkdir -p $D
for f in bench_c1 bench_c1_s121 bench_c4_s20w5 bench_c2 bench_c3 bench_c3_s20w5 bench_c4 bench_c5 bench_c3_K32 bench_c3_K33 bench_c3_K36 bench_c3_K40 bench_c3_K44 bench_c3_K47 bench_c3_K48 bench_c3_K63 bench_c3_ctns2 grid_c1_k4 grid_c2_k4 grid_c3_k2 prof_c3 prof_c5; do
  [ -f $S/$f.json ] && cp $S/$f.json $D/
done
cp $(ls -t $S/prof_c3/*/*kernel_stats.csv | head -1) $D/c3_kernel_stats.csv
cp $(ls -t $S/prof_c5/*/*kernel_stats.csv | head -1) $D/c5_kernel_stats.csv
cp $S/c3_steady_iteration_timeline.txt $S/c3_trace_avg.txt $S/slab8.log $S/slab8_timeline.txt $S/concurrent_grids.log $D/ 2>/dev/null
cp $S/c3_pmc_issue.csv $D/c3_pmc_issue.csv
cp $S/c3_K40_pmc_issue.csv $D/c3_K40_pmc_issue.csv 2>/dev/null
cp $S/c5_pmc_issue.csv $D/c5_pmc_issue.csv 2>/dev/null
cp $S/issue.json $R/profiles/issue.json
# (gpurun MERGES a call's outputs into gpurun_out/: a second profile job of the round leaves the first one's counter files beside
# its own, and the traffic figures would be sums over both runs — keep the newest file of each pass only)
for d in $S/pmc_fetch $S/pmc_write; do
  for sub in $d/*/; do
    for kind in counter_collection agent_info; do
      ls -t $sub*_$kind.csv 2>/dev/null | tail -n +2 | while read f; do rm -f "$f"; done
    done
  done
done
python3 $R/tools/pmc_csv.py $S/pmc_fetch $S/pmc_write > $D/c3_pmc_summary.csv
cd $R && INSIDER_COMMIT=${INSIDER_COMMIT:-$(git rev-parse --short HEAD)} python3 tools/pmc_traffic.py c3 $S/pmc_fetch $S/pmc_write $(git rev-parse --short HEAD)
python3 - <<PY
import json
out = {}
for key, f, cmd in (("c4", "bench_c4.json", "--steps 11 --warmup 1"), ("c4_s20w5", "bench_c4_s20w5.json", "--steps 20 --warmup 5")):
    try:
        d = json.loads(open("$S/" + f).readline())
    except Exception:
        continue
    out[key] = {"n_gpus": 1, "value": d["value"], "unit": d["unit"], "steps": d["steps"], "warmup": d["warmup"], "ms_per_step": d["ms_per_step"],
                "source": "profiles/$TAG/" + f + " (python bench.py --workload c4 " + cmd + ", one MI355X, this repository's run)"}
if out:
    json.dump(out, open("$R/profiles/single_gpu.json", "w"), indent=1)
PY
ls $D
