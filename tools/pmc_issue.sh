#!/bin/bash
# Issue-slot counters (SQ block, 8 slots per pass) for the kernels of a c3 call: three rocprofv3 --pmc passes of the same
# command, merged by tools/pmc_issue.py into gpurun_out/$1/c3_pmc_issue.csv + .json.  Counters only: no trace domain.
set -o pipefail
TAG=${1:-r04}
WL=${2:-c3}
EXTRA=${3:-}          # further bench.py arguments (e.g. "--latent 40")
LBL=${4:-$WL}         # label of the outputs and of the entry in issue.json
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 -c "import sys; sys.path.insert(0, '$R'); import __graft_entry__ as g; g.build()" || exit 1
CMD="python3 $R/bench.py --workload $WL --steps 4 --warmup 0 --no-cpu-baseline $EXTRA"
P1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"
P2="SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VALU SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_BRANCH SQ_IFETCH GRBM_GUI_ACTIVE"
P3="SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_BUSY_CU_CYCLES SQ_CYCLES GRBM_GUI_ACTIVE"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i + 1))
  rocprofv3 --pmc $P --output-format csv -d $OUT/pmc_issue_${LBL}_$i -- $CMD > $OUT/pmc_issue_${LBL}_$i.json 2> $OUT/pmc_issue_${LBL}_$i.err || { tail -5 $OUT/pmc_issue_${LBL}_$i.err; exit 1; }
  echo "pass $i done"
done
cd $R
INSIDER_PMC_EXTRA="$EXTRA" INSIDER_ISSUE_JSON=$OUT/issue.json python3 tools/pmc_issue.py $LBL $OUT/${LBL}_pmc_issue $OUT/pmc_issue_${LBL}_1 $OUT/pmc_issue_${LBL}_2 $OUT/pmc_issue_${LBL}_3
echo PMC_ISSUE_DONE
