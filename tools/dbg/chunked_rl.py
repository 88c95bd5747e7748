"""Large batches: does running the relational layer CHUNK BY CHUNK (conv stack and question encoder on the whole batch, autograd then
runs backward chain -> reductions -> weight gradient per chunk, so a chunk's dZ is read back while it still sits in the Infinity
Cache) beat one launch sequence over the whole batch?  Captured trainer steps, dropout 0, un-fused loss in both arms.
    python tools/dbg/chunked_rl.py --batch 640 --chunks 0 64 128 160 320"""
import argparse, json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=640)
    ap.add_argument("--chunks", type=int, nargs="+", default=[0, 128])
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--config", default="original-fp")
    args = ap.parse_args()
    import bench
    import relationnetworks_clevr_amd as pkg
    from relationnetworks_clevr_amd import dp
    dev = torch.device("cuda", 0)
    hyp = json.load(open(os.path.join(ROOT, "relationnetworks-clevr_amd", "config.json")))["hyperparams"][args.config]
    hyp = dict(hyp, dropout=0.0)
    B = args.batch
    img, qst, lab = bench.make_batch(B, dev, 128, state_desc=bool(hyp["state_description"]))
    ref = None
    for rep in range(2):
        for chunk in args.chunks:
            with pkg.options.override(fused_loss=False):
                torch.manual_seed(42)
                model = bench.quiet_rn(pkg, dict(hyp))
                model.cuda(dev)
                model.train()
                if chunk:
                    orig = model.rl.forward

                    def chunked(x, q, label=None, coord=None, _o=orig, _c=chunk):
                        outs = [_o(x[i:i + _c], q[i:i + _c], coord=coord) if coord is not None else _o(x[i:i + _c], q[i:i + _c])
                                for i in range(0, x.shape[0], _c)]
                        return torch.cat(outs)
                    model.rl.forward = chunked
                opt = torch.optim.Adam(model.parameters(), lr=5e-6, weight_decay=1e-4)
                tr = dp.DataParallelTrainer(model, opt, clip_norm=50.0, use_graph=True, copy_guard_every=0)
                bufs = tr.input_buffers(img, qst, lab)
                for _ in range(5):
                    loss = tr.step(*bufs)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    loss = tr.step(*bufs)
                torch.cuda.synchronize()
                ms = 1e3 * (time.perf_counter() - t0) / args.steps
                print("B %d chunk %4d: %.3f ms / step  %.1f k q/s   loss %.6f" % (B, chunk, ms, B / ms, float(loss.detach())), flush=True)
                del tr, model, opt
                torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
