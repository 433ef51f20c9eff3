#!/bin/bash
# re-run single cases of a fuzz_parity.py sweep under alternative kernel forms: tools/fuzz_repro.sh <cases> <seed> <case> ...
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
N=$1; SEED=$2; shift 2
for c in "$@"; do
  for o in "" "col_factored=0,row_merged=0" "cd_variant=1" "cd_variant=2" "cd_pass1=0" "row_counts=0,col_factored=2"; do
    FUZZ_ONLY=$c FUZZ_OPTS=$o python tests/fuzz_parity.py $N $SEED 2>&1 | grep -E "^case|MISMATCH" | cut -c1-260
  done
done
