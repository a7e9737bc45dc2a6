"""Kernels around the step boundary of a rocprofv3 kernel trace (the optimiser launch of the second-to-last step and
what follows it):  python tools/trace_boundary.py <kernel_trace.csv> [before] [after]"""
import csv, gzip, re, sys
op = gzip.open if sys.argv[1].endswith(".gz") else open
rows = list(csv.DictReader(op(sys.argv[1], "rt")))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Stream_Id"], r["Queue_Id"]) for r in rows)
before, after = int(sys.argv[2]) if len(sys.argv) > 2 else 6, int(sys.argv[3]) if len(sys.argv) > 3 else 40
def short(n):
    n = re.sub(r"\(anonymous namespace\)::|_ZN12_GLOBAL__N_1\d+", "", n)
    m = re.search(r"(\w+_kernel|\w+Functor|copyBuffer|fillBuffer\w*)", n)
    return (m.group(1) if m else n)[:34]
adam = [i for i, e in enumerate(ev) if "adam_multi" in e[2]]
i0 = adam[-2]
t0 = ev[i0][0]
end = max(e[1] for e in ev[:max(1, i0 - before)])
for s, e, n, st, q in ev[i0 - before:i0 + after]:
    print("%9.1f us  +%7.1f us  gap %7.1f  stream %s queue %s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - end) / 1e3, st, q, short(n)))
    end = max(end, e)
