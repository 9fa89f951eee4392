#!/bin/bash
# rocprofv3 passes of the fused cross-attention kernel at the headline shape, ON THE GPU BOX from the repo root: bash tools/profile_ca.sh r03
# kernel-trace stats, SQ / FETCH / WRITE counters in SEPARATE passes (MI355X_MICROARCH.md section HBM).  Outputs under gpurun_out/<tag>_ca_prof/.
set -o pipefail
TAG=${1:-r03}
MODE=${2:-mixed}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${TAG}_ca_prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/tools/prof_ca.py $MODE 12"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $CMD > $OUT/trace.log 2>&1 && echo "trace ok"
rocprofv3 --kernel-trace --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d $OUT/pmc_sq -o p -- $CMD > $OUT/pmc_sq.log 2>&1 && echo "pmc sq ok"
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/pmc_fetch -o p -- $CMD > $OUT/pmc_fetch.log 2>&1 && echo "pmc fetch ok"
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT/pmc_write -o p -- $CMD > $OUT/pmc_write.log 2>&1 && echo "pmc write ok"
cd $R && python3 tools/pmc_summary.py $OUT/pmc_summary.json $OUT/pmc_sq $OUT/pmc_fetch $OUT/pmc_write
cp $OUT/trace/*kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null
ls $OUT
