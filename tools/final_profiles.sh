# Regenerates the measurement artefacts of the default workload in one GPU session (outputs under gpurun_out/final/):
#   bash tools/final_profiles.sh
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final
mkdir -p $O
cd $R
timeout -k 10 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout -k 10 300 python bench.py --all-kernel-timing --no-cpu-baseline --per-shape --per-shape-top 90 > $O/bench_all.json 2> $O/per_shape.txt
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/bench_steps20_warmup5.json 2> $O/bench_steps20_warmup5.err
timeout -k 10 300 python bench.py --launch graph --no-cpu-baseline > $O/bench_graph.json 2> $O/bench_graph.err
timeout -k 10 300 python bench.py --launch eager --no-cpu-baseline > $O/bench_eager.json 2> $O/bench_eager.err
timeout -k 10 300 python bench.py --ddp-single --launch graph --no-cpu-baseline > $O/bench_ddp_single_graph.json 2> $O/bench_ddp_single_graph.err
timeout -k 10 300 python bench.py --ddp-single --no-cpu-baseline > $O/bench_ddp_single.json 2> $O/bench_ddp_single.err
bash tools/ab_bench.sh URESNET_WIDE 2 > $O/ab_wide.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o p -- python3 $R/bench.py --launch eager --no-cpu-baseline > $O/bench_prof.json 2> $O/bench_prof.err
# the same with everything on one stream: the per-kernel durations of `roofline` (the run above is host-bound under the profiler,
# so its two streams hardly overlap and its durations agree with these)
URESNET_BRANCH_STREAMS=0 timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof1 -o p1 -- python3 $R/bench.py --launch eager --no-cpu-baseline > $O/bench_prof1.json 2> $O/bench_prof1.err
rm -f $O/prof/*kernel_trace.csv $O/prof1/*kernel_trace.csv
cd $R && bash tools/pmc_bench.sh
# then, here:  P=profiles/rNN_x; python tools/pmc_traffic.py gpurun_out/pmc_fetch/f_counter_collection.csv gpurun_out/pmc_write/w_counter_collection.csv ${P}_pmc_traffic.json
#              and copy gpurun_out/final/{bench_*.json, per_shape.txt, ab_wide.txt, prof/p_kernel_stats.csv, prof1/p1_kernel_stats.csv} to ${P}_*
