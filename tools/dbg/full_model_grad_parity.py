#!/usr/bin/env python3
"""Round 6, the ir-fp plateau lag: every parameter gradient of the FULL model (conv + question encoder + relational layer) in the
default arithmetic against the fp32 mode of this package, same weights, same batch -- at initialisation and ON the plateau of the
relational task (after N fp32 steps), for ir-fp and, as the control, original-fp.  A term that is wrong or mis-scaled outside the
relational layer's own fixtures (the question gradient's way back into the encoder, the conv grid's gradient) would show here.
usage: full_model_grad_parity.py [steps_on_plateau=600]"""
import contextlib, io, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
import relationnetworks_clevr_amd as pkg
from relationnetworks_clevr_amd import train as T, dp

N = int(sys.argv[1]) if len(sys.argv) > 1 else 600
hyps = json.load(open(os.path.join(ROOT, "relationnetworks-clevr_amd", "config.json")))["hyperparams"]


class A:
    qdict_size, adict_size = 82, 28


def model(name, prec, state=None, seed=0):
    torch.manual_seed(seed)
    with contextlib.redirect_stdout(io.StringIO()):
        m = pkg.RN(A, dict(hyps[name], precision=prec, dropout=0.0))
    m.cuda(); m.train()
    if state is not None:
        m.load_state_dict(state)
    return m


def grads(m, batch):
    for p in m.parameters():
        p.grad = None
    img, qst, lab = batch
    out = m(img, qst)
    loss = torch.nn.functional.nll_loss(out, lab)
    loss.backward()
    torch.cuda.synchronize()
    return float(loss), {k: p.grad.detach().double().clone() for k, p in m.named_parameters() if p.grad is not None}


for name in ("ir-fp", "original-fp"):
    data = iter(T.PairRelationTaskOnDevice(N + 4, 64, seed=11, device="cuda"))
    m32 = model(name, "fp32")
    for when in ("at initialisation", "after %d fp32 steps (on the plateau)" % N):
        if when.startswith("after"):
            opt = torch.optim.Adam(m32.parameters(), lr=5e-4, weight_decay=1e-4)
            tr = dp.DataParallelTrainer(m32, opt, clip_norm=50.0, use_graph=False)
            for _ in range(N):
                last = float(tr.step(*next(data)))
            print("   (fp32 loss after %d steps: %.3f)" % (N, last))
        m32.train(); m32.conv.eval()                       # (running statistics: the two modes then see the same normalisation)
        batch = next(data)
        state = {k: v.clone() for k, v in m32.state_dict().items()}
        l32, g32 = grads(m32, batch)
        ma = model(name, "auto", state); ma.conv.eval()
        la, ga = grads(ma, batch)
        print("== %s, %s: loss fp32 %.6f auto %.6f" % (name, when, l32, la))
        worst = []
        for k in g32:
            a, b = ga[k], g32[k]
            rel = float((a - b).norm() / b.norm().clamp_min(1e-30))
            cos = float((a * b).sum() / (a.norm() * b.norm()).clamp_min(1e-30))
            ratio = float(a.norm() / b.norm().clamp_min(1e-30))
            worst.append((rel, k, cos, ratio, float(b.norm())))
        for rel, k, cos, ratio, nb in sorted(worst, reverse=True)[:12]:
            print("   %-28s rel L2 %.3e  cos %.6f  |auto|/|fp32| %.4f  |fp32| %.3e" % (k, rel, cos, ratio, nb))
        m32.train()
