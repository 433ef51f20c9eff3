#!/bin/bash
# A/B of the row phase's level Gram sums: k_wsyrk (row_gemm=0) against k_wgemm at several slab counts
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for o in "row_gemm=0" "row_gemm_waves=512" "row_gemm_waves=1024" "row_gemm_waves=2048"; do
echo "== $o"
bash tools/quick_bench.sh --opt $o
done
