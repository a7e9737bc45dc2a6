# A/B of one environment switch of bench.py inside one GPU session:  bash tools/ab_bench.sh VAR [reps] [extra bench args]
# prints value, ms per step and the dominant kernel's numbers for VAR=0 and VAR=1, alternating.
VAR=$1; REPS=${2:-2}; shift; shift
for rep in $(seq $REPS); do
  for f in 0 1; do
    env $VAR=$f timeout -k 10 300 python bench.py --no-cpu-baseline --steps 12 "$@" > /tmp/ab_$f.json 2>/dev/null || exit 1
    python - "$VAR" $f <<'PY'
import json, sys
d = json.load(open("/tmp/ab_%s.json" % sys.argv[2]))
r = d["roofline"] or {"achieved": 0.0, "frac": 0.0, "avg_launch_ms": 0.0}          # null under --no-kernel-timing
print("%s=%s  %.2f images/s  %.3f ms/step  dominant %.1f TF frac %.4f avg launch %.4f ms" % (
    sys.argv[1], sys.argv[2], d["value"], d["ms_per_step"], r["achieved"], r["frac"], r["avg_launch_ms"]), flush=True)
PY
  done
done
