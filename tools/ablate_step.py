"""Critical-path attribution for the graph-replayed training step.

rocprofv3's kernel trace serialises the queues, so it cannot say which kernels sit on the critical
path of the multi-stream step.  This tool removes one C-ABI wrapper at a time (replaced by a no-op
before the hipGraph is captured; outputs stay uninitialised, which does not matter for timing) and
reports how much the step gets shorter: that is the wrapper's critical-path contribution.

    python tools/ablate_step.py [--precision bf16] [--batch 64] [--only name,name]"""
import argparse
import contextlib
import io
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import make_batch                                                        # noqa: E402

KEEP = ("load", "tile", "bytes", "ok", "supported", "pack", "available", "chunk")   # query / setup helpers stay


def build(pkg, dp, hyp, dev, B, hw):
    class A:
        qdict_size, adict_size = 82, 28
    torch.manual_seed(42)
    with contextlib.redirect_stdout(io.StringIO()):
        model = pkg.RN(A, hyp)
    model.cuda(dev).train()
    opt = torch.optim.Adam(model.parameters(), lr=5e-6, weight_decay=1e-4, fused=True)
    return dp.DataParallelTrainer(model, opt, clip_norm=50.0, use_graph=True)


def time_step(tr, batch, steps=30, warm=6):
    for _ in range(warm):
        tr.step(*batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.step(*batch)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e6


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--precision", default="auto")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--hw", type=int, default=128)
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    import relationnetworks_clevr_amd as pkg
    from relationnetworks_clevr_amd import dp
    H = pkg.rn_hip
    H.load()
    dev = torch.device("cuda", 0)
    hyps = json.load(open(os.path.join(ROOT, "relationnetworks-clevr_amd", "config.json")))["hyperparams"]
    hyp = dict(hyps["original-fp"], precision=args.precision)
    batch = make_batch(args.batch, dev, args.hw)
    base = time_step(build(pkg, dp, hyp, dev, args.batch, args.hw), batch)
    print(f"baseline step {base:8.1f} us")
    names = [n for n in dir(H) if callable(getattr(H, n)) and not n.startswith("_") and n.islower()
             and getattr(getattr(H, n), "__module__", "") == H.__name__ and not any(k in n for k in KEEP)]
    if args.only:
        names = [n for n in names if n in args.only.split(",")]
    rows = []
    for n in names:
        orig = getattr(H, n)
        setattr(H, n, lambda *a, **k: None)
        try:
            t = time_step(build(pkg, dp, hyp, dev, args.batch, args.hw), batch, steps=20, warm=4)
            rows.append((base - t, n))
        except Exception as e:                                                      # wrapper's return value is needed
            rows.append((float("nan"), n + "  (" + type(e).__name__ + ")"))
        finally:
            setattr(H, n, orig)
    for d, n in sorted(rows, key=lambda r: -(r[0] if r[0] == r[0] else -1e9)):
        print(f"{d:8.1f} us  {n}")


if __name__ == "__main__":
    main()
