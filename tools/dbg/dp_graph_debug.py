"""debug: gradients of graph-replayed steps on the B=8 batch vs its two B=4 shards (no process group)"""
import sys, os, io, contextlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np, torch
import relationnetworks_clevr_amd as pkg
from relationnetworks_clevr_amd import dp
from oracle import formula
class A: qdict_size, adict_size = 82, 28
x = torch.from_numpy(formula.hash_uniform((8, 3, 128, 128), 5, 0.0, 1.0)).cuda()
q = torch.from_numpy(formula.hash_ints((8, 12), 6, 1, 83)).cuda()
y = torch.from_numpy(formula.hash_ints((8,), 7, 0, 28)).cuda()
def grads(sl, use_graph, conv_eval=True):
    torch.manual_seed(3)
    with contextlib.redirect_stdout(io.StringIO()):
        m = pkg.RN(A, dict(formula.HYP["original-fp"], dropout=0.0))
    m.cuda(); m.train()
    if conv_eval: m.conv.eval()
    opt = torch.optim.Adam(m.parameters(), lr=0.0, eps=1e-1)
    tr = dp.DataParallelTrainer(m, opt, clip_norm=None, use_graph=use_graph)
    tr._fused_opt = None
    class NoOpt:
        def step(self): pass
    tr.opt = NoOpt()
    out = []
    for _ in range(3):
        loss = tr.step(x[sl].contiguous(), q[sl].contiguous(), y[sl].contiguous())
        torch.cuda.synchronize()
        out.append(tr.bucket.flat.clone())
    names = [(n, p.numel()) for n, p in m.named_parameters() if p.requires_grad]
    return out, names, float(loss)
for conv_eval in (True, False):
    for ug in (False, True):
        full, names, lf = grads(slice(0, 8), ug, conv_eval); a, _, la = grads(slice(0, 4), ug, conv_eval); b, _, lb = grads(slice(4, 8), ug, conv_eval)
        print("conv_eval", conv_eval, "graph", ug, "loss full %.6f shards %.6f" % (lf, 0.5 * (la + lb)), "replays identical:", torch.equal(full[0], full[2]), torch.equal(a[0], a[2]))
        if conv_eval:
            s = 0.5 * (a[2] + b[2]); off = 0; worst = []
            for n, k in names:
                e = float((s[off:off + k] - full[2][off:off + k]).norm() / max(float(full[2][off:off + k].norm()), 1e-30)); worst.append((e, n)); off += k
            worst.sort(reverse=True); print("   worst:", [("%.2e" % e, n) for e, n in worst[:4]])
