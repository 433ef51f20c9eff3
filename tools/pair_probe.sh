#!/bin/bash
# Timing-only experiment (profiles/r04/exp_quick/pair_probe.patch): the K = 30 sweep kernel with TWO coordinate steps per computed
# jump (wrong order; a sweep of 30 jumps makes 60 steps) against the shipped one, on the fixed-sweep-count probe of
# tools/cd_probe.py.  The pair builds' "ns per wave-step" figures count 30 steps per sweep: halve them.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for n in default pair pair8 default; do
  if [ $n = default ]; then unset INSIDER_HIP_LIB; else export INSIDER_HIP_LIB=$R/tools/_ab/lib_$n.so; fi
  echo "== $n"
  python3 tools/cd_probe.py probe 30 2>&1 | grep "B="
done
echo PAIR_DONE
