#!/bin/bash
# Long randomised parity sweeps on the library in the tree (one gpurun call).  Outputs under gpurun_out/fuzz_long.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/fuzz_long
mkdir -p $OUT
cd $R
python3 -c "import __graft_entry__ as g; g.build()" || exit 1
timeout -k 10 120 python3 -m pytest tests/test_gpu_parity.py -x -q -k "inplace_update" 2>&1 | tail -2
FUZZ_DUMP=$OUT/dump_a.jsonl timeout -k 10 420 python3 tests/fuzz_parity.py 3500 777001 > $OUT/fuzz_parity_3500_777001.txt 2>&1; echo "fuzz_parity rc $?"; tail -1 $OUT/fuzz_parity_3500_777001.txt
FUZZ_K=33,34,36,37,40,41,44,45,47,48 FUZZ_DUMP=$OUT/dump_b.jsonl timeout -k 10 200 python3 tests/fuzz_parity.py 800 777002 > $OUT/fuzz_parity_reg3_800_777002.txt 2>&1; echo "fuzz_parity reg3 rc $?"; tail -1 $OUT/fuzz_parity_reg3_800_777002.txt
timeout -k 10 200 python3 tests/fuzz_cd.py 6000 777003 > $OUT/fuzz_cd_6000_777003.txt 2>&1; echo "fuzz_cd rc $?"; tail -1 $OUT/fuzz_cd_6000_777003.txt
echo FUZZ_DONE
