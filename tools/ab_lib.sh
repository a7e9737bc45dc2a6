# A/B of two library builds on the default workload, interleaved: bash tools/ab_lib.sh <other.so> [rounds]
R=$GRAFT_REPO_ROOT; cd $R
for i in $(seq 1 ${2:-3}); do
  for v in base other; do
    if [ $v = other ]; then export URESNET_HIP_LIB=$R/$1; else unset URESNET_HIP_LIB; fi
    timeout -k 10 200 python bench.py --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'])" || exit 1
  done
done
