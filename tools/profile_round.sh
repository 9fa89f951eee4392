#!/bin/bash
# rocprofv3 passes of one round, run ON THE GPU BOX from the repo root:  bash tools/profile_round.sh r02
# kernel-trace stats of the default bench command, SQ / FETCH / WRITE counters in SEPARATE passes (MI355X_MICROARCH.md section HBM),
# kernel trace of the voxelisers.  Outputs under gpurun_out/<tag>_prof/; summaries are copied into profiles/ by hand.
set -o pipefail
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${TAG}_prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --no-extras --no-cpu-baseline --steps 3 --warmup 2"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -o b -- $BENCH > $OUT/bench.log 2>&1 && echo "bench trace ok"
rocprofv3 --kernel-trace --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d $OUT/pmc_sq -o p -- $BENCH > $OUT/pmc_sq.log 2>&1 && echo "pmc sq ok"
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/pmc_fetch -o p -- $BENCH > $OUT/pmc_fetch.log 2>&1 && echo "pmc fetch ok"
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT/pmc_write -o p -- $BENCH > $OUT/pmc_write.log 2>&1 && echo "pmc write ok"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/vox -o v -- python3 $R/tools/prof_voxel.py > $OUT/vox.log 2>&1 && echo "voxel trace ok"
cd $R && python3 tools/pmc_summary.py $OUT/pmc_summary.json $OUT/pmc_sq $OUT/pmc_fetch $OUT/pmc_write
ls -la $OUT
