"""debug: does an eager allocation between graph replays change the training result? (use-after-free hunt)"""
import sys, os, io, contextlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np, torch
import relationnetworks_clevr_amd as pkg
from relationnetworks_clevr_amd import dp
import relationnetworks_clevr_amd.train as T
from oracle import formula
class A: qdict_size, adict_size = 82, 28

def run(with_eval, junk_sizes=(), fill=0.0, seed=0, steps=5):
    torch.manual_seed(seed)
    with contextlib.redirect_stdout(io.StringIO()):
        m = pkg.RN(A, dict(formula.HYP["original-fp"], dropout=0.0)).cuda()
    opt = torch.optim.Adam(m.parameters(), lr=1e-4, weight_decay=1e-4)
    tr = dp.DataParallelTrainer(m, opt, clip_norm=50.0, use_graph=True)
    batch = next(iter(T.SyntheticClevr(8, 8, seed=3)))
    img, qq, yy = T.load_tensor_data(batch, "cuda")
    losses = []
    for it in range(steps):
        m.train()
        losses.append(float(tr.step(img, qq, yy).detach()))
        if with_eval:
            m.eval()
            with torch.no_grad():
                for b in (3, 5, 16):
                    m(torch.rand(b, 3, 128, 128, device="cuda"), torch.randint(1, 83, (b, 20), device="cuda"))
        junk = [torch.full((sz,), fill, device="cuda") for sz in junk_sizes for _ in range(64)]
        torch.cuda.synchronize()
        del junk
    return losses
base = run(False)
print("base          ", base)
for name, kw in [("eval", dict(with_eval=True)), ("junk1k", dict(with_eval=False, junk_sizes=(1024,), fill=float("nan"))),
                 ("eval+junk1k", dict(with_eval=True, junk_sizes=(1024,), fill=float("nan"))),
                 ("eval+junk many", dict(with_eval=True, junk_sizes=(16, 128, 1024, 8192, 65536, 1 << 20), fill=float("nan")))]:
    r = run(**kw)
    print("%-14s" % name, r, "SAME" if r == base else "DIFFERENT")
