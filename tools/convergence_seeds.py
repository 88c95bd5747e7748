"""Does the default arithmetic train like the reference's fp32?  (reference loop: train.py:36-48.)

Several seeds x {fp32, auto (e4m3 copies), auto with 16-bit copies} x {original-fp, ir-fp} on the relational synthetic task
(train.PairRelationTaskOnDevice: batches made on the device, ~3 s a run instead of ~2 min), one JSON line per run into
gpurun_out/convergence_seeds.jsonl, and a summary table (mean +- s.d. per cell) at the end:

    python tools/convergence_seeds.py [--seeds 8] [--steps 3000] [--lr 5e-4] [--models original-fp,ir-fp] [--modes fp32,auto,auto16]

Per run: `exit_step` = first step at which the trailing 250-step mean loss is below 0.6 (the task's plateau: the one-object
"which column" questions are solved, the pair questions are not, sits at ~0.95), `final_loss` = mean of the last 250 steps,
`accuracy` on 32 held-out batches in eval mode.  A seed fixes the initial weights, the batches and the cell draws for every mode."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np

MODES = {"fp32": ("fp32", None), "auto": ("auto", True), "auto16": ("auto", False), "bf16": ("bf16", True)}


def exit_step(curve, every, window=250, thr=0.6):
    k = max(window // every, 1)
    for i in range(k, len(curve) + 1):
        if float(np.mean(curve[i - k:i])) < thr:
            return i * every
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=8)
    ap.add_argument("--first-seed", type=int, default=0)
    ap.add_argument("--steps", type=int, default=3000)
    ap.add_argument("--lr", type=float, default=5e-4)
    ap.add_argument("--every", type=int, default=50)
    ap.add_argument("--models", default="original-fp,ir-fp")
    ap.add_argument("--modes", default="fp32,auto,auto16")
    ap.add_argument("--eval-batches", type=int, default=32)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "convergence_seeds.jsonl"))
    ap.add_argument("--summary-of", default=None, metavar="JSONL", help="no runs: print the summary tables of an earlier --out file")
    ap.add_argument("--override", default="", help="bisection arms: options.OPT overrides for every run, e.g. chain_reduce=0,fphi_split=0")
    ap.add_argument("--sched", default="", help="... and functional.SCHED knobs, e.g. dq_async=0")
    ap.add_argument("--tag", default="", help="suffix of the mode name in the output rows (one arm = one tag)")
    ap.add_argument("--dither", type=int, default=0, help="bisection arm: tile-dithered weight images per layer of the forward chain (1, 2, 4 = product, 8)")
    a = ap.parse_args()
    if a.summary_of:
        rows = [json.loads(ln) for ln in open(a.summary_of) if ln.strip()]
        print("# %d runs of %s: %s" % (len(rows), os.path.basename(a.summary_of), sorted({(r["model"], r["mode"]) for r in rows})))
        return summarize(rows, sorted({r["model"] for r in rows}, reverse=True), [m for m in MODES if any(r["mode"] == m for r in rows)])
    import relationnetworks_clevr_amd as pkg                # (imported here: --summary-of and the tests' use of exit_step need no GPU)
    from relationnetworks_clevr_amd import train as T
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    import contextlib
    ov = {k: (v not in ("0", "False", "false")) for k, v in (kv.split("=") for kv in a.override.split(",") if kv)}
    for kv in a.sched.split(","):
        if kv:
            k, v = kv.split("=")
            pkg.functional.SCHED[k] = int(v)
    if a.dither:
        pkg.rn_hip.F16S_DITHER = a.dither
    rows = []
    with open(a.out, "a") as f:
        for model in a.models.split(","):
            for seed in range(a.first_seed, a.first_seed + a.seeds):
                for mode in a.modes.split(","):
                    prec, h8 = MODES[mode]
                    t0 = time.time()
                    with (pkg.options.override(**ov) if ov else contextlib.nullcontext()):
                        r = T.convergence_run(prec, steps=a.steps, lr=a.lr, h8=h8, task="pairs_dev", log_every=a.every,
                                              eval_batches=a.eval_batches, model_name=model, seed=seed)
                    k = max(250 // a.every, 1)
                    row = {"model": model, "mode": mode + a.tag, "seed": seed, "steps": a.steps, "lr": a.lr, "exit_step": exit_step(r["loss"], a.every),
                           "final_loss": float(np.mean(r["loss"][-k:])), "accuracy": r["accuracy"], "seconds": round(time.time() - t0, 1),
                           "switched": any(g.get("switched") for g in r["copy_guard"]), "every": a.every,
                           "loss": [round(v, 4) for v in r["loss"]]}
                    rows.append(row)
                    f.write(json.dumps(row) + "\n"); f.flush()
                    import gc, torch
                    gc.collect()
                    print("%-11s %-6s seed %d: exit %s final %.4f acc %.4f (%.1f s; %.1f GB allocated)" % (
                        model, mode, seed, row["exit_step"], row["final_loss"], row["accuracy"], row["seconds"], torch.cuda.memory_allocated() / 1e9), flush=True)
    summarize(rows, a.models.split(","), a.modes.split(","))


def summarize(rows, models, modes):
    print("\n# summary: mean +- s.d. over seeds (exit_step over the runs that left the plateau; `stuck` = runs that did not within --steps)")
    print("%-11s %-6s %2s  %-16s %-5s %-18s %-16s" % ("model", "mode", "n", "exit_step", "stuck", "final_loss", "accuracy"))
    for model in models:
        for mode in modes:
            c = [r for r in rows if r["model"] == model and r["mode"] == mode]
            if not c:
                continue
            ex = [r["exit_step"] for r in c if r["exit_step"] is not None]
            fl, ac = [r["final_loss"] for r in c], [r["accuracy"] for r in c]
            print("%-11s %-6s %2d  %7.0f +- %-5.0f %-5d %.4f +- %-8.4f %.4f +- %.4f" % (
                model, mode, len(c), np.mean(ex) if ex else float("nan"), np.std(ex) if ex else float("nan"), len(c) - len(ex),
                np.mean(fl), np.std(fl), np.mean(ac), np.std(ac)))
    print("\n# paired differences to fp32, same seed (= same initial weights, same batches): mean, s.d., standard error, t = mean / s.e.")
    for model in models:
        by = {(r["mode"], r["seed"]): r for r in rows if r["model"] == model}
        seeds = sorted({s_ for (_m, s_) in by})
        for mode in modes:
            if mode == "fp32":
                continue
            pairs = [(by[(mode, s_)], by[("fp32", s_)]) for s_ in seeds if (mode, s_) in by and ("fp32", s_) in by]
            if len(pairs) < 2:
                continue
            for key in ("accuracy", "final_loss", "exit_step"):
                d = np.array([a_[key] - b_[key] for a_, b_ in pairs if a_[key] is not None and b_[key] is not None], dtype=np.float64)
                se = d.std(ddof=1) / np.sqrt(len(d))
                print("%-11s %-6s - fp32  %-10s n %2d  mean %+9.4f  s.d. %8.4f  s.e. %8.4f  t %+5.2f" % (model, mode, key, len(d), d.mean(), d.std(ddof=1), se, d.mean() / se))


if __name__ == "__main__":
    main()
