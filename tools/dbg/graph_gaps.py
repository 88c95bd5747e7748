#!/usr/bin/env python3
"""Idle time inside a captured step: from a rocprofv3 kernel trace (rocpd .db or csv) of `bench.py` in hipGraph mode, take the
last steady-state steps and list the intervals in which no kernel runs at all (what the graph's launch latency costs)."""
import sys, sqlite3, csv
path = sys.argv[1]
if path.endswith(".db"):
    db = sqlite3.connect(path)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table' or type='view'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    ev = [(s, e, n) for s, e, n in db.execute("select d.start, d.end, k.kernel_name from %s d join %s k on d.kernel_id = k.id order by d.start" % (kd, ks))]
else:
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(path)))
marks = [s for s, e, n in ev if "pair_tables_kernel" in n]
nst = int(sys.argv[2]) if len(sys.argv) > 2 else 5
t0, t1 = marks[-nst - 1], marks[-1]
win = [(s, e, n) for s, e, n in ev if t0 <= s < t1]
idle, gaps, cur_e, last = 0, [], None, None
for s, e, n in win:
    if cur_e is not None and s > cur_e:
        idle += s - cur_e
        gaps.append((s - cur_e, last, n))
    if cur_e is None or e > cur_e:
        cur_e, last = e, n
print("%d steps: %.1f us/step, idle (no kernel running) %.1f us/step in %d gaps/step" % (nst, (t1 - t0) / 1e3 / nst, idle / 1e3 / nst, len(gaps) // nst))
import collections
agg = collections.defaultdict(lambda: [0, 0])
for g, a, b in gaps:
    k = (a[:40], b[:40]); agg[k][0] += g; agg[k][1] += 1
for (a, b), (g, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:45]:
    print("  %6.2f us x%.0f  %-40s -> %s" % (g / 1e3 / c, c / nst, a, b))
if len(sys.argv) > 3:
    a, b = marks[-2], marks[-1]
    one = [(s, e, n) for s, e, n in ev if a - 200000 <= s < b - 200000]
    base = one[0][0]
    print("one step, kernels in start order (start us, duration us):")
    for s, e, n in one:
        print("  %8.1f %7.1f  %s" % ((s - base) / 1e3, (e - s) / 1e3, n[:110]))
