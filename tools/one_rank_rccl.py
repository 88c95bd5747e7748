"""What a one-GPU box can execute of the N > 1 step over RCCL: a ONE-rank `nccl` communicator and
DataParallelTrainer(single_rank_exchange=True) -- start-up self-check of a captured RCCL all-reduce, capture of forward + backward +
all-reduce + clip / Adam into one hipGraph, replay -- timed beside the plain one-rank step (no collective) and beside the eager
exchange (all-reduce launched behind the replayed forward + backward).  The sum over one rank is the identity: the three runs must
leave the same parameters.  Prints one JSON line.   python tools/one_rank_rccl.py [--steps 200] [--batch 64]"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--config", default="original-fp")
    args = ap.parse_args()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    import bench
    import relationnetworks_clevr_amd as pkg
    from relationnetworks_clevr_amd import dp
    hyp = json.load(open(os.path.join(ROOT, "relationnetworks-clevr_amd", "config.json")))["hyperparams"][args.config]
    img, qst, lab = bench.make_batch(args.batch, dev, 128, state_desc=bool(hyp["state_description"]))
    out, ref = {}, None
    for name, kw in (("plain", {}), ("in_graph", {"single_rank_exchange": True}),
                     ("eager_exchange", {"single_rank_exchange": True, "graph_allreduce": False})):
        torch.manual_seed(42)
        model = bench.quiet_rn(pkg, dict(hyp))
        model.cuda(dev)
        model.train()
        opt = torch.optim.Adam(model.parameters(), lr=5e-6, weight_decay=1e-4)
        tr = dp.DataParallelTrainer(model, opt, clip_norm=50.0, use_graph=True, copy_guard_every=0, **kw)
        bufs = tr.input_buffers(img, qst, lab)
        for _ in range(10):
            tr.step(*bufs)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            tr.step(*bufs)
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / args.steps
        flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
        if ref is None:
            ref = flat.clone()
        out[name] = {"ms_per_step": round(ms, 4), "exchange_mode": tr.exchange_mode(), "exchange_fallback": tr.exchange_fallback,
                     "checks": tr.exchange_checks, "params_equal_plain": bool(torch.equal(flat, ref))}
    out["allreduce_bytes"] = 4 * tr.bucket.numel
    print(json.dumps(out))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
