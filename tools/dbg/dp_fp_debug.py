"""debug: (1) gradient of the B=8 batch vs the two B=4 shards, per parameter, in f16s / fp32; (2) run-to-run determinism of the graph trainer"""
import sys, os, io, contextlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import relationnetworks_clevr_amd as pkg
from relationnetworks_clevr_amd import dp
import relationnetworks_clevr_amd.train as T
from oracle import formula

class A: qdict_size, adict_size = 82, 28

def model(prec, seed=3):
    torch.manual_seed(seed)
    with contextlib.redirect_stdout(io.StringIO()):
        m = pkg.RN(A, dict(formula.HYP["original-fp"], dropout=0.0, precision=prec))
    m.cuda(); m.train(); m.conv.eval()
    return m

x = torch.from_numpy(formula.hash_uniform((8, 3, 128, 128), 5, 0.0, 1.0)).cuda()
q = torch.from_numpy(formula.hash_ints((8, 12), 6, 1, 83)).cuda()
y = torch.from_numpy(formula.hash_ints((8,), 7, 0, 28)).cuda()

def grads(m, sl):
    for p in m.parameters(): p.grad = None
    out = m(x[sl].contiguous(), q[sl].contiguous())
    torch.nn.functional.nll_loss(out, y[sl], reduction="sum").backward()
    torch.cuda.synchronize()
    return {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}

for prec in ("fp32", "f16s", "bf16"):
    m = model(prec)
    gfull = grads(m, slice(0, 8)); ga = grads(m, slice(0, 4)); gb = grads(m, slice(4, 8))
    worst = []
    for n in gfull:
        s = ga[n] + gb[n]
        e = float((s - gfull[n]).norm() / max(float(gfull[n].norm()), 1e-30))
        worst.append((e, n))
    worst.sort(reverse=True)
    print(prec, "shard-sum vs full, worst:", [(("%.2e" % e), n) for e, n in worst[:5]])
    if prec == "fp32":
        ref = gfull
    else:
        w2 = sorted(((float((gfull[n] - ref[n]).norm() / max(float(ref[n].norm()), 1e-30)), n) for n in gfull), reverse=True)
        print(prec, "full vs fp32 full, worst:", [(("%.2e" % e), n) for e, n in w2[:5]])
        w3 = sorted(((float((ga[n] + gb[n] - ref[n]).norm() / max(float(ref[n].norm()), 1e-30)), n) for n in gfull), reverse=True)
        print(prec, "shard-sum vs fp32 full, worst:", [(("%.2e" % e), n) for e, n in w3[:5]])
        g0 = gfull["rl.g_layers.0.weight"]; r0 = ref["rl.g_layers.0.weight"]; s0 = (ga["rl.g_layers.0.weight"] + gb["rl.g_layers.0.weight"])
        for nm, c0, c1 in (("x_j", 0, 26), ("x_i", 26, 52), ("q", 52, 180)):
            print("   W0 cols", nm, "full err %.2e  shard err %.2e" % (float((g0[:, c0:c1] - r0[:, c0:c1]).norm() / r0[:, c0:c1].norm()), float((s0[:, c0:c1] - r0[:, c0:c1]).norm() / r0[:, c0:c1].norm())))

def run(with_eval, seed=0):
    torch.manual_seed(seed)
    with contextlib.redirect_stdout(io.StringIO()):
        m = pkg.RN(A, dict(formula.HYP["original-fp"], dropout=0.0)).cuda()
    opt = torch.optim.Adam(m.parameters(), lr=1e-4, weight_decay=1e-4)
    tr = dp.DataParallelTrainer(m, opt, clip_norm=50.0, use_graph=True)
    batch = next(iter(T.SyntheticClevr(8, 8, seed=3)))
    img, qq, yy = T.load_tensor_data(batch, "cuda")
    losses = []
    for it in range(6):
        m.train()
        losses.append(float(tr.step(img, qq, yy).detach()))
        if with_eval:
            m.eval()
            with torch.no_grad():
                for b in (3, 5, 16):
                    m(torch.rand(b, 3, 128, 128, device="cuda"), torch.randint(1, 83, (b, 20), device="cuda"))
    return losses
print("no eval  :", run(False)); print("no eval  :", run(False)); print("with eval:", run(True)); print("with eval:", run(True))
