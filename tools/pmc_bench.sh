# HBM traffic of every kernel of the default bench workload, from the L2's memory-side counters.
# Two passes (FETCH_SIZE takes 3 of the 4 TCC counters, WRITE_SIZE 2):   bash tools/pmc_bench.sh
# Output: gpurun_out/pmc_fetch/, gpurun_out/pmc_write/ (csv); tools/pmc_traffic.py turns them into profiles/*.json
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --launch eager"
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_fetch -o f -- python3 $R/bench.py $ARGS > $R/gpurun_out/pmc_fetch.log 2>&1
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_write -o w -- python3 $R/bench.py $ARGS > $R/gpurun_out/pmc_write.log 2>&1
