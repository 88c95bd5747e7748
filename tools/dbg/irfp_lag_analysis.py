#!/usr/bin/env python3
"""Paired statistics of the ir-fp bisection arms (tools/dbg/irfp_lag_bisect.sh) against the fp32 and the default-mode runs of the
same seeds (profiles/r06_convergence_seeds.jsonl: seeds 0-23, r06_convergence_seeds_irfp_seeds24_47.jsonl: 24-47)."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
P = lambda f: os.path.join(ROOT, "profiles", f)
base = [json.loads(l) for f in ("r06_convergence_seeds.jsonl", "r06_convergence_seeds_irfp_seeds24_47.jsonl") for l in open(P(f))]
arms = [json.loads(l) for l in open(sys.argv[1] if len(sys.argv) > 1 else P("r06_convergence_seeds_irfp_bisect.jsonl"))]
MODEL = sys.argv[2] if len(sys.argv) > 2 else "ir-fp"
ref = {m: {r["seed"]: r for r in base if r["model"] == MODEL and r["mode"] == m} for m in ("fp32", "auto", "auto16")}
rows = {"auto (default)": ref["auto"], "auto16": ref["auto16"]}
for r in arms:
    if r["model"] == MODEL:
        rows.setdefault(r["mode"], {})[r["seed"]] = r
print("# " + MODEL + ", 3000 steps: each arm against fp32 on the same seed (mean +- s.e., t); exit_step over the pairs in which both left the plateau")
print("%-24s %3s  %-26s %-26s %-8s" % ("arm", "n", "exit_step - fp32", "accuracy - fp32", "stuck"))
for name, d in rows.items():
    out = []
    for key in ("exit_step", "accuracy"):
        x = np.array([d[s][key] - ref["fp32"][s][key] for s in sorted(d) if s in ref["fp32"] and d[s][key] is not None and ref["fp32"][s][key] is not None], float)
        if len(x) < 2:
            out.append("-")
            continue
        se = x.std(ddof=1) / np.sqrt(len(x))
        out.append("%+8.3f +- %-7.3f (t %+.2f)" % (x.mean(), se, x.mean() / se) if key == "accuracy" else "%+6.0f +- %-4.0f (t %+.2f)" % (x.mean(), se, x.mean() / se))
    print("%-24s %3d  %-26s %-26s %d" % (name, len(d), out[0], out[1], sum(1 for s in d if d[s]["exit_step"] is None)))
print("fp32: %d of %d runs still on the plateau at 3000" % (sum(1 for s in ref["fp32"] if ref["fp32"][s]["exit_step"] is None), len(ref["fp32"])))
