#!/bin/bash
# rocprofv3 passes of the dynamic voxeliser alone (cfg-3: 8 x 65 536 points, 0.1 m grid, 3-D keys), ON THE GPU BOX: bash tools/profile_voxel_dyn.sh r03
set -o pipefail
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${TAG}_dyn_prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/tools/prof_voxel_dyn.py"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $CMD > $OUT/trace.log 2>&1 && echo "trace ok"
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/pmc_fetch -o p -- $CMD > $OUT/pmc_fetch.log 2>&1 && echo "pmc fetch ok"
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT/pmc_write -o p -- $CMD > $OUT/pmc_write.log 2>&1 && echo "pmc write ok"
cd $R && python3 tools/pmc_summary.py $OUT/pmc_summary.json $OUT/pmc_fetch $OUT/pmc_write
cp $OUT/trace/*kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null
