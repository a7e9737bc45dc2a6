# exact read bytes of the tile kernel per launch from the L2's memory-side request counters by size:
#   bash tools/pmc_tile_bytes.sh <lib.so> <shape index of tools/bench_tile.py> [dgrad_bn]
# bytes = 32 * RDREQ_32B + 64 * RDREQ_64B + 128 * RDREQ_128B (FETCH_SIZE folds the sizes with a formula that tallies 128-byte
# requests only through TCC_BUBBLE)
set -e
LIB=$(realpath $1); export TILE_SHAPE=$2; [ -n "$3" ] && export TILE_MODE=$3
R=$GRAFT_REPO_ROOT; D=$R/gpurun_out/pmc_bytes_$2_${3:-fwd}
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --kernel-trace --output-format csv -d $D -- python3 $R/tools/bench_tile.py $LIB > $D.log 2>&1
python3 - $D <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    if "conv3x3_tile_kernel" in r["Kernel_Name"]:
        acc[r["Kernel_Name"][:60]][r["Counter_Name"]] += float(r["Counter_Value"]); n[(r["Kernel_Name"][:60], r["Counter_Name"])] += 1
for k, c in acc.items():
    l = max(v for (kk, _), v in n.items() if kk == k)
    a, b, d = (c.get("TCC_EA0_RDREQ_%s_sum" % s, 0.0) / l for s in ("32B", "64B", "128B"))
    print("%s: %d launches, per launch 32B %.3g  64B %.3g  128B %.3g requests = %.1f MB read" % (k, l, a, b, d, (32 * a + 64 * b + 128 * d) / 1e6))
PY
