#!/bin/bash
# The REAL blocks of two coordinate steps (profiles/r04/exp_quick/pair_blocks.patch built as tools/_ab/lib_pairs.so) on the
# fixed-sweep-count probe of tools/cd_probe.py (identical problems that never converge: every wave full), with pairs of a whole
# slot, of lane groups of 8 / 4, and without pairs (order_mode's bits 8..), against the shipped library.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
echo "== shipped"; python3 tools/cd_probe.py probe 30 2>&1 | grep "B="
export INSIDER_HIP_LIB=$R/tools/_ab/lib_pairs.so
for OM in 0 2048 1024 256; do
  echo "== pair build, order_mode $OM"; PROBE_ORDER_MODE=$OM python3 tools/cd_probe.py probe 30 2>&1 | grep "B="
done
unset INSIDER_HIP_LIB
echo "== shipped"; python3 tools/cd_probe.py probe 30 2>&1 | grep "B="
echo PAIR2_DONE
