# Matrix-pipe and LDS counters of the three 3x3 kernels on the 128 -> 128 @ 256x512 layer, one SQ pass and one GRBM pass per
# program:  bash tools/pmc_sq.sh   (outputs: gpurun_out/pmc_sq_{wide,wgrad}_{a,b}/ as csv; tools/pmc_sq_summary.py reads them)
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp WD_SHAPE=0 WG_SHAPE=0 URESNET_WGRAD_V=4
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"
timeout -s KILL 120 rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d $R/gpurun_out/pmc_sq_wide_a -o a -- python3 $R/tools/bench_wide.py > $R/gpurun_out/pmc_sq_wide_a.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --kernel-trace --output-format csv -d $R/gpurun_out/pmc_sq_wide_b -o b -- python3 $R/tools/bench_wide.py > $R/gpurun_out/pmc_sq_wide_b.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d $R/gpurun_out/pmc_sq_wgrad_a -o a -- python3 $R/tools/bench_wgrad.py > $R/gpurun_out/pmc_sq_wgrad_a.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --kernel-trace --output-format csv -d $R/gpurun_out/pmc_sq_wgrad_b -o b -- python3 $R/tools/bench_wgrad.py > $R/gpurun_out/pmc_sq_wgrad_b.log 2>&1
