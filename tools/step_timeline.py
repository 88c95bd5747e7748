"""Concurrent timeline of the graph-replayed training step from in-graph clock stamps.

Every C-ABI wrapper of rn_hip (and the trainer's fwd+bwd as a whole) is bracketed by one-thread stamp kernels
(rn_debug_stamp) on the stream it is launched on; the stamps are captured into the hipGraph with everything else,
so after a replay the buffer holds begin / end device times of every op with the streams running concurrently.
The stamps cost a few microseconds per op: read the table for structure (what overlaps what, which stream ends
last), not for absolute step time.

    python tools/step_timeline.py [--precision bf16] [--batch 64] [--config ir-fp]"""
import argparse
import contextlib
import io
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import make_batch                                                        # noqa: E402

SKIP = ("load", "tile", "bytes", "ok", "supported", "available", "chunk", "debug_stamp", "dtype")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--precision", default="auto")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--hw", type=int, default=128)
    ap.add_argument("--config", default="original-fp")
    args = ap.parse_args()
    import relationnetworks_clevr_amd as pkg
    from relationnetworks_clevr_amd import dp
    H = pkg.rn_hip
    H.load()
    dev = torch.device("cuda", 0)
    hyps = json.load(open(os.path.join(ROOT, "relationnetworks-clevr_amd", "config.json")))["hyperparams"]
    hyp = dict(hyps[args.config], precision=args.precision)

    class A:
        qdict_size, adict_size = 82, 28
    torch.manual_seed(42)
    with contextlib.redirect_stdout(io.StringIO()):
        model = pkg.RN(A, hyp)
    model.cuda(dev).train()
    opt = torch.optim.Adam(model.parameters(), lr=5e-6, weight_decay=1e-4, fused=True)
    tr = dp.DataParallelTrainer(model, opt, clip_norm=50.0, use_graph=True)
    batch = make_batch(args.batch, dev, args.hw, state_desc=bool(hyp.get("state_description")))

    buf = torch.zeros(8192, dtype=torch.int64, device=dev)
    state = {"idx": 0, "ops": []}
    streams = {}

    def mark(name, phase):
        i = state["idx"]
        state["idx"] += 1
        sid = streams.setdefault(torch.cuda.current_stream().cuda_stream, len(streams))
        state["ops"].append((i, name, phase, sid))
        H.debug_stamp(buf, i)

    def wrap(name, fn):
        def w(*a, **k):
            mark(name, "b")
            r = fn(*a, **k)
            mark(name, "e")
            return r
        return w

    for n in dir(H):
        f = getattr(H, n)
        if callable(f) and not n.startswith("_") and n.islower() and getattr(f, "__module__", "") == H.__name__ \
                and not any(k in n for k in SKIP):
            setattr(H, n, wrap(n, f))
    orig = tr._fwd_bwd

    def fwd_bwd(*a):
        state["idx"] = 0
        state["ops"] = []
        mark("STEP", "b")
        r = orig(*a)
        mark("STEP", "e")
        return r
    tr._fwd_bwd = fwd_bwd
    for _ in range(6):
        tr.step(*batch)
    torch.cuda.synchronize()
    t = buf.cpu().numpy()
    ops = state["ops"]
    t0 = t[ops[0][0]]
    span = t[ops[-1][0]] - t0
    tick = 0.01                                        # wall_clock64: 100 MHz
    print(f"step span {span * tick:8.1f} us ({len(ops) // 2} bracketed ops, {len(streams)} streams)")
    begins = {}
    rows = []
    for i, name, ph, sid in ops:
        if ph == "b":
            begins[(name, sid)] = t[i]
        else:
            b = begins.pop((name, sid))
            rows.append(((b - t0) * tick, (t[i] - t0) * tick, sid, name))
    print(f"{'begin':>8} {'end':>8} {'dur':>7}  s  op")
    for b, e, sid, name in sorted(rows):
        print(f"{b:8.1f} {e:8.1f} {e - b:7.1f}  {sid}  {'  ' * sid}{name}")


if __name__ == "__main__":
    main()
