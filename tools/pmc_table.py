#!/usr/bin/env python3
"""Per-kernel mean of every PMC counter in a rocprofv3 counter_collection.csv (+ derived ratios)."""
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
import os, sys as _s
_s.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kname import pretty
for k, d in acc.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    name = pretty(k)
    line = "%-96s" % name
    wc = m.get("SQ_WAVE_CYCLES", 0)
    for c in sorted(m):
        line += " %s=%.3g" % (c.replace("SQ_", ""), m[c])
    if wc:
        line += " | wait_any/wave=%.2f inst_wait/wave=%.2f active/wave=%.2f" % (
            m.get("SQ_WAIT_ANY", 0) / wc, m.get("SQ_WAIT_INST_ANY", 0) / wc, m.get("SQ_ACTIVE_INST_ANY", 0) / wc)
        if m.get("SQ_BUSY_CYCLES"):
            line += " mfma_busy/busy=%.3f" % (m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / m["SQ_BUSY_CYCLES"])
        if m.get("SQ_LDS_IDX_ACTIVE"):
            line += " bankconf/lds_active=%.3f" % (m.get("SQ_LDS_BANK_CONFLICT", 0) / m["SQ_LDS_IDX_ACTIVE"])
    print(line)
