#!/bin/bash
# A/B of library options under the kernel-trace profiler (steady-state means, tools/iter_timeline.py): tools/ab_opts.sh "opt=v" ...
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/ab
mkdir -p $OUT
cd $R
i=0
for rep in 1 2; do
for o in "profile=1" "$@"; do
  i=$((i + 1))
  D=$OUT/opt_$i
  rm -rf $D
  (cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --output-format csv -d $D -- python3 $R/bench.py --no-cpu-baseline --opt $o > $D.json 2> $D.err)
  python3 tools/iter_timeline.py $D > $D.timeline.txt 2>&1
  echo "$o: $(python3 -c "import json; print('value %.1f' % json.loads(open('$D.json').readline())['value'])") $(tail -1 $D.timeline.txt | cut -d: -f2-)"
  rm -rf $D
done
done
