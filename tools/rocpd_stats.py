#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite database (kernel trace) as a per-kernel table (CSV-ish).
usage: rocpd_stats.py results.db [steps]  -> name, calls, total_ms, avg_us, pct"""
import sqlite3, sys
db = sys.argv[1]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else None
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
rows = c.execute("select name, count(*), sum(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
print("kernel,calls,total_ms,avg_us,min_us,max_us,pct" + (",ms_per_step" if steps else ""))
for name, n, t, mn, mx in rows:
    short = name if len(name) < 110 else name[:107] + "..."
    line = '"%s",%d,%.3f,%.2f,%.2f,%.2f,%.2f' % (short, n, t / 1e6, t / n / 1e3, mn / 1e3, mx / 1e3, 100.0 * t / tot)
    if steps:
        line += ",%.4f" % (t / 1e6 / steps)
    print(line)
print('"TOTAL",%d,%.3f,,,,100' % (sum(r[1] for r in rows), tot / 1e6))
