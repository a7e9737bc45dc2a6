"""Idle gaps between consecutive kernels of a rocprofv3 kernel trace:  python tools/trace_gaps.py <kernel_trace.csv> [steps]"""
import csv, re, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); m = re.search(r"(\w+_kernel|\w+Functor|copyBuffer)", n); return m.group(1) if m else n[:30]
# last third of the trace = steady state
ev = ev[len(ev) * 2 // 3:]
span = ev[-1][1] - ev[0][0]; busy = sum(e - s for s, e, _ in ev)
print("kernels %d  span %.2f ms  busy %.2f ms  idle %.2f ms (%.1f %%)" % (len(ev), span / 1e6, busy / 1e6, (span - busy) / 1e6, 100 * (span - busy) / span))
gaps = collections.defaultdict(lambda: [0, 0])
big = []
end = ev[0][1]
for (s, e, n), (ps, pe, pn) in zip(ev[1:], ev[:-1]):
    g = s - max(end, pe); end = max(end, e)
    if g > 0:
        k = (short(pn), short(n)); gaps[k][0] += 1; gaps[k][1] += g
        if g > 20000: big.append((g, short(pn), short(n)))
for k, (c, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:15]:
    print("%8.1f us total %5d x %6.2f us   %s -> %s" % (t / 1e3, c, t / c / 1e3, k[0], k[1]))
print("gaps > 20 us:", sorted(big, reverse=True)[:10])
