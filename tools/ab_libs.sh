#!/bin/bash
# A/B of several builds on ONE box: tools/ab_libs.sh OUTDIR ROUNDS "bench args" tag1 tag2 ...   (tag "tree" = the library in the
# tree, any other tag = tools/_ab/lib_<tag>.so, a build with other -D flags or of another commit); one summary line per run
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1; ROUNDS=$2; ARGS=$3; shift 3
mkdir -p $OUT
cd $R
line() {
python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=((d['roofline'].get('steady_state') or {}).get('avg_launch_ms_parts') or {'sweeps':float('nan'),'statistics':float('nan')})
print('$1: value %.1f ms/step %.3f | cd %.3f ms stats %.4f | steady cd %.3f stats %.3f | G updates/s %.1f | loss %.12g | sha %s'%(d['value'],d['ms_per_step'],d['cd_kernel']['avg_launch_ms'],d['masked_gram']['avg_launch_ms'],r['sweeps'],r['statistics'],d['cd_kernel']['coordinate_updates_per_s']/1e9,d['loss'],d.get('library_source_sha')))"
}
for r in $(seq 1 $ROUNDS); do
  for t in "$@"; do
    if [ $t = tree ]; then unset INSIDER_HIP_LIB; else export INSIDER_HIP_LIB=$R/tools/_ab/lib_$t.so; fi
    python bench.py --no-cpu-baseline $ARGS 2>>$OUT/err_$t.log | tee $OUT/bench_${t}_$r.json | line $t
  done
done
echo AB_LIBS_DONE
