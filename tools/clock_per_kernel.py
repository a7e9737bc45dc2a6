"""Average shader clock per kernel from a rocprofv3 database made with `--pmc GRBM_GUI_ACTIVE --kernel-trace`:
active cycles of a dispatch / its duration.  python tools/clock_per_kernel.py results.db [min_us]"""
import collections
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    min_ns = float(sys.argv[2]) * 1e3 if len(sys.argv) > 2 else 1e5
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    t = {k: next(n for n in tabs if n.startswith(k)) for k in ("rocpd_kernel_dispatch_", "rocpd_info_kernel_symbol_", "rocpd_pmc_event_")}
    q = ("select s.kernel_name, d.start, d.end, e.value from %s d join %s s on d.kernel_id = s.id join %s e on e.event_id = d.event_id"
         % (t["rocpd_kernel_dispatch_"], t["rocpd_info_kernel_symbol_"], t["rocpd_pmc_event_"]))
    agg = collections.defaultdict(list)
    for name, s, e, v in cur.execute(q):
        if e - s >= min_ns:                       # the counter has a fixed offset that dominates short dispatches
            agg[name].append((e - s, v))
    rows = []
    for name, l in agg.items():
        dur, cyc = sum(x[0] for x in l), sum(x[1] for x in l)
        short = re.sub(r"_ZN12_GLOBAL__N_1\d+|\(anonymous namespace\)::|\.kd$", "", name)
        rows.append((dur, "%-64s launches %5d  avg %7.1f us  %.2f GHz" % (short[:64], len(l), dur / len(l) / 1e3, cyc / dur)))
    for _, line in sorted(rows, reverse=True):
        print(line)


if __name__ == "__main__":
    main()
