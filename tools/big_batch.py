"""The reference's DEFAULT batch size is 640 (train.py:370-372).  One batch of B questions against the same questions in chunks of 64:
log-probs (eval, the batch-invariant arithmetic -> tight; training arithmetic -> 1e-3 contract) and parameter gradients of the
relational layer (conv in eval mode so that the chunks are independent).  python tools/big_batch.py [--batch 640]"""
import argparse, json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(batch=640, config="original-fp", precision="auto"):
    args = argparse.Namespace(batch=batch, config=config, precision=precision)
    import bench
    import relationnetworks_clevr_amd as pkg
    dev = torch.device("cuda", 0)
    hyp = json.load(open(os.path.join(ROOT, "relationnetworks-clevr_amd", "config.json")))["hyperparams"][args.config]
    hyp = dict(hyp, dropout=0.0, precision=args.precision)
    B = args.batch
    img, qst, lab = bench.make_batch(B, dev, 128, state_desc=bool(hyp["state_description"]))
    torch.manual_seed(42)
    model = bench.quiet_rn(pkg, hyp)
    model.cuda(dev)
    out = {"batch": B, "config": args.config, "precision": args.precision}
    # eval
    model.eval()
    with torch.no_grad():
        big = model(img, qst)
        parts = torch.cat([model(img[i:i + 64], qst[i:i + 64]) for i in range(0, B, 64)])
    torch.cuda.synchronize()
    out["eval_logprob_max_abs_diff"] = float((big - parts).abs().max())
    out["eval_argmax_agree"] = float((big.argmax(1) == parts.argmax(1)).float().mean())
    out["finite"] = bool(torch.isfinite(big).all())
    # training arithmetic, gradients (sum-reduced loss so that chunks add up)
    model.train()
    if hasattr(model, "conv"):
        model.conv.eval()
    def grads(chunks):
        model.zero_grad(set_to_none=True)
        tot = 0.0
        for i, j in chunks:
            o = model(img[i:j], qst[i:j])
            l = torch.nn.functional.nll_loss(o, lab[i:j], reduction="sum")
            l.backward()
            tot += float(l.detach())
        torch.cuda.synchronize()
        return tot, {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
    lb, gb = grads([(0, B)])
    lp, gp = grads([(i, min(i + 64, B)) for i in range(0, B, 64)])
    out["train_loss_sum_big_vs_chunks"] = [lb, lp, abs(lb - lp) / abs(lp)]
    worst = {}
    for n in gp:
        if n.startswith("rl.") or n.startswith("text."):
            e = float((gb[n] - gp[n]).norm() / (gp[n].norm() + 1e-30))
            worst[n] = e
    out["grad_rel_l2_big_vs_chunks_max"] = max(worst.values())
    out["grad_rel_l2_worst"] = sorted(worst.items(), key=lambda kv: -kv[1])[:4]
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=640)
    ap.add_argument("--config", default="original-fp")
    ap.add_argument("--precision", default="auto")
    a = ap.parse_args()
    print(json.dumps(run(a.batch, a.config, a.precision)))
