"""Is the captured step's gradient bitwise reproducible from replay to replay?  lr = 0 and weight_decay = 0 freeze the parameters
(conv in eval mode: BatchNorm buffers frozen too), so every replay must leave the SAME flat gradient bucket.  Prints, per parameter
tensor, how many of N replays differ from the first and the largest relative difference.
    python tools/dbg/grad_repro.py [--replays 200] [--batch 8] [--eager] [--config original-fp]"""
import argparse, json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--replays", type=int, default=200)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--config", default="original-fp")
    ap.add_argument("--eager", action="store_true")
    ap.add_argument("--bn-train", action="store_true")
    ap.add_argument("--new-trainer-every", type=int, default=0, help="rebuild model + trainer (same seed) every this many replays")
    args = ap.parse_args()
    import bench
    import relationnetworks_clevr_amd as pkg
    from relationnetworks_clevr_amd import dp
    dev = torch.device("cuda", 0)
    hyp = json.load(open(os.path.join(ROOT, "relationnetworks-clevr_amd", "config.json")))["hyperparams"][args.config]
    hyp = dict(hyp, dropout=0.0)
    img, qst, lab = bench.make_batch(args.batch, dev, 128, state_desc=bool(hyp["state_description"]))

    def build():
        torch.manual_seed(42)
        model = bench.quiet_rn(pkg, dict(hyp))
        model.cuda(dev)
        model.train()
        if not args.bn_train and hasattr(model, "conv"):
            model.conv.eval()
        opt = torch.optim.Adam(model.parameters(), lr=0.0, weight_decay=0.0)
        tr = dp.DataParallelTrainer(model, opt, clip_norm=50.0, use_graph=not args.eager, copy_guard_every=0)
        return model, tr

    model, tr = build()
    names = [n for n, p in model.named_parameters() if p.requires_grad]
    ref, ndiff, worst = None, {}, {}
    for r in range(args.replays):
        if args.new_trainer_every and r and r % args.new_trainer_every == 0:
            model, tr = build()
        tr.step(img, qst, lab)
        torch.cuda.synchronize()
        flat = tr.bucket.flat.clone()
        if ref is None:
            ref = flat
            continue
        if torch.equal(flat, ref):
            continue
        for nme, o, p in zip(names, tr.bucket.offsets, tr.bucket.params):
            a, b = flat[o:o + p.numel()], ref[o:o + p.numel()]
            if not torch.equal(a, b):
                ndiff[nme] = ndiff.get(nme, 0) + 1
                rel = float((a - b).norm() / (b.norm() + 1e-30))
                worst[nme] = max(worst.get(nme, 0.0), rel)
    print("replays %d, eager %s, batch %d: tensors that differed from replay 0:" % (args.replays, args.eager, args.batch))
    for nme in names:
        if nme in ndiff:
            print("  %-32s %4d replays   worst rel L2 diff %.3e" % (nme, ndiff[nme], worst[nme]))
    if not ndiff:
        print("  none (bitwise reproducible)")


if __name__ == "__main__":
    main()
