#!/usr/bin/env python3
"""From a rocprofv3 kernel_trace.csv: per-step busy time, union-of-intervals time and the biggest kernels,
for the steady-state window (skips the first `skip` fraction of the trace)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
# steady state: last `steps` occurrences window delimited by the FusedAdam kernel (one per step)
marks = [s for s, e, n in ev if "pair_build_kernel" in n]
if len(marks) > steps + 1:
    t0, t1 = marks[-steps - 1], marks[-1]
else:
    t0, t1 = ev[len(ev) // 2][0], ev[-1][1]
win = [(s, e, n) for s, e, n in ev if s >= t0 and s < t1]
busy = sum(e - s for s, e, _ in win)
union, cur_s, cur_e = 0, None, None
for s, e, _ in win:
    if cur_e is None or s > cur_e:
        if cur_e is not None: union += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
union += cur_e - cur_s
n = max(1, min(steps, len(marks) - 1))
print("window %.3f ms for %d steps -> %.3f ms/step; sum of kernel time %.3f ms/step; GPU non-idle (union) %.3f ms/step" % (
    (t1 - t0) / 1e6, n, (t1 - t0) / 1e6 / n, busy / 1e6 / n, union / 1e6 / n))
agg = collections.defaultdict(lambda: [0, 0])
for s, e, nme in win:
    import re as _re
    key = _re.sub(r"^void ", "", nme)
    key = key.replace("at::native::", "").replace("(anonymous namespace)::", "")
    a = agg[key[:100]]; a[0] += e - s; a[1] += 1
for k, (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:int(sys.argv[3]) if len(sys.argv) > 3 else 28]:
    print("  %-100s %7.1f us/step  x%d" % (k, t / 1e3 / n, c // n))
