# hardware-counter passes over tools/bench_tile.py for one build of the tile kernel: bash tools/pmc_tile.sh <lib.so>
set -e
LIB=$(realpath $1)
cd /tmp && export TMPDIR=/tmp TILE_SHAPE=${TILE_SHAPE:-0}
R=$GRAFT_REPO_ROOT
timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/pmcF -- python3 $R/tools/bench_tile.py $LIB > $R/gpurun_out/pmcF.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --kernel-trace --output-format csv -d $R/gpurun_out/pmcG -- python3 $R/tools/bench_tile.py $LIB > $R/gpurun_out/pmcG.log 2>&1
