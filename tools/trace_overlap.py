"""How much two streams overlap in a rocprofv3 kernel trace:  python tools/trace_overlap.py <kernel_trace.csv>
For the last third of the trace (steady state): wall span, time with at least one / at least two kernels running,
sum of kernel durations, kernels per hardware queue."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?")) for r in rows)
ev = ev[len(ev) * 2 // 3:]
points = []
for s, e, _ in ev:
    points.append((s, 1))
    points.append((e, -1))
points.sort()
depth, last, one, two = 0, points[0][0], 0, 0
for t, d in points:
    if depth >= 1:
        one += t - last
    if depth >= 2:
        two += t - last
    depth += d
    last = t
span = ev[-1][1] - ev[0][0]
queues = collections.Counter(q for _, _, q in ev)
print("kernels %d  span %.2f ms  busy(>=1) %.2f ms  overlapped(>=2) %.2f ms  sum of durations %.2f ms  idle %.2f ms  queues %s"
      % (len(ev), span / 1e6, one / 1e6, two / 1e6, sum(e - s for s, e, _ in ev) / 1e6, (span - one) / 1e6, dict(queues)))
