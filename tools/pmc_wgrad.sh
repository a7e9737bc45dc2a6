# hardware-counter passes over tools/bench_wgrad.py: bash tools/pmc_wgrad.sh <tag>   (URESNET_WGRAD_V, WG_SHAPE from the environment)
set -e
TAG=$1
cd /tmp && export TMPDIR=/tmp WG_SHAPE=${WG_SHAPE:-0}
R=$GRAFT_REPO_ROOT
timeout -s KILL 120 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $R/gpurun_out/pmcW${TAG}a -- python3 $R/tools/bench_wgrad.py > $R/gpurun_out/pmcW${TAG}a.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d $R/gpurun_out/pmcW${TAG}b -- python3 $R/tools/bench_wgrad.py > $R/gpurun_out/pmcW${TAG}b.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmcW${TAG}c -- python3 $R/tools/bench_wgrad.py > $R/gpurun_out/pmcW${TAG}c.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum --kernel-trace --output-format csv -d $R/gpurun_out/pmcW${TAG}d -- python3 $R/tools/bench_wgrad.py > $R/gpurun_out/pmcW${TAG}d.log 2>&1
