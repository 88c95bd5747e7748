#!/usr/bin/env python3
"""Splits the gradient error of the f16s mode on fixture G-fp64 (GPU): the module's dx / dq against the fixture, and the SAME
quantities from an fp32 torch backward that uses the ReLU gates the forward kernel actually produced (its lane masks) -- i.e. the
part of the error that comes from flipped gates alone -- plus the per-layer fraction of gates that differ from an fp32 forward's."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import gold
import relationnetworks_clevr_amd as pkg
from oracle import formula
from test_gpu_kernels import rr_mask_decode

F = pkg.functional
cap = {}
orig = F.chain_forward
def wrap(*a, **k):
    r = orig(*a, **k); cap["masks"] = r[1]; return r
F.chain_forward = wrap

def l2rel(a, ref):
    a = np.asarray(a, np.float64); ref = np.asarray(ref, np.float64)
    return float(np.linalg.norm(a - ref) / np.linalg.norm(ref))

tag = sys.argv[1] if len(sys.argv) > 1 else "G-fp64"
g = gold.load(tag)
hyp, sd, x, q, lab = gold.rl_case(g["meta"])
B, n, k = x.shape; Q = q.shape[1]; M = B * n * n
rl = pkg.RelationalLayer(hyp["rl_in_size"], formula.ADICT, hyp["lstm_hidden"], dict(hyp, precision="f16s"))
rl.load_state_dict({k_: torch.from_numpy(v) for k_, v in sd.items()}, strict=True)
rl = rl.cuda().eval()
xt = torch.from_numpy(x).cuda().requires_grad_(True); qt = torch.from_numpy(q).cuda().requires_grad_(True)
lp = rl(xt, qt)
torch.nn.functional.nll_loss(lp, torch.from_numpy(lab).cuda()).backward()
torch.cuda.synchronize()
print("module (f16s):                      dx %.2e  dq %.2e" % (l2rel(xt.grad.cpu().numpy(), g["dx"]), l2rel(qt.grad.cpu().numpy(), g["dq"])))
masks = cap["masks"].masks
gates = [torch.from_numpy(rr_mask_decode(masks[l], M, l)).cuda() for l in range(4)]
W = [torch.from_numpy(sd["g_layers.%d.weight" % l]).cuda() for l in range(4)]
b = [torch.from_numpy(sd["g_layers.%d.bias" % l]).cuda() for l in range(4)]
fW = [torch.from_numpy(sd["f_fc%d.weight" % (i + 1)]).cuda() for i in range(3)]
fb = [torch.from_numpy(sd["f_fc%d.bias" % (i + 1)]).cuda() for i in range(3)]
torch.backends.cuda.matmul.allow_tf32 = False

def pairs(xt, qt):
    xj = xt[:, None, :, :].expand(B, n, n, k); xi = xt[:, :, None, :].expand(B, n, n, k); qq = qt[:, None, None, :].expand(B, n, n, Q)
    return torch.cat([xj, xi, qq], 3).reshape(M, 2 * k + Q)

def run(gs, xg_noise=0.0):
    x2 = torch.from_numpy(x).cuda().requires_grad_(True); q2 = torch.from_numpy(q).cuda().requires_grad_(True)
    h = pairs(x2, q2)
    ref_g = []
    bs = [t.clone().requires_grad_(True) for t in b]
    for l in range(4):
        z = h @ W[l].t() + bs[l]
        ref_g.append(z.detach() > 0)
        h = z * (gs[l] if gs is not None else ref_g[l])
    xg = h.view(B, n * n, -1).sum(1)
    if xg_noise:
        torch.manual_seed(5); xg = xg * (1.0 + xg_noise * torch.randn_like(xg))
    f = torch.relu(xg @ fW[0].t() + fb[0]); f = torch.relu(f @ fW[1].t() + fb[1])
    torch.nn.functional.nll_loss(torch.log_softmax(f @ fW[2].t() + fb[2], 1), torch.from_numpy(lab).cuda()).backward()
    run.db = [t.grad.cpu().numpy() for t in bs]
    return x2.grad.cpu().numpy(), q2.grad.cpu().numpy(), ref_g

dx0, dq0, ref_g = run(None)
print("fp32 torch on this GPU, own gates:  dx %.2e  dq %.2e" % (l2rel(dx0, g["dx"]), l2rel(dq0, g["dq"])))
dx1, dq1, _ = run(gates)
print("fp32 torch, the KERNEL's gates:     dx %.2e  dq %.2e   gates that differ from the fp32 forward's, per layer: %s" % (
    l2rel(dx1, g["dx"]), l2rel(dq1, g["dq"]), " ".join("%.1e" % float((a != r).float().mean()) for a, r in zip(gates, ref_g))))
print("module vs fp32-with-kernel-gates:   dx %.2e  dq %.2e   (what the backward kernels' own arithmetic adds)" % (
    l2rel(xt.grad.cpu().numpy(), dx1), l2rel(qt.grad.cpu().numpy(), dq1)))
mod_db = [rl.g_layers[l].bias.grad.cpu().numpy() for l in range(4)]
print("bias gradients per g layer, module vs fixture:               " + " ".join("%.2e" % l2rel(mod_db[l], g["grad/g_layers.%d.bias" % l]) for l in range(4)))
print("bias gradients per g layer, module vs fp32-with-kernel-gates: " + " ".join("%.2e" % l2rel(mod_db[l], run.db[l]) for l in range(4)))
oh = np.zeros_like(g["log_probs"]); oh[np.arange(B), lab] = 1.0
dl_mod = np.exp(lp.detach().cpu().numpy()) - oh; dl_ref = np.exp(g["log_probs"]) - oh
print("d loss / d logits = softmax - onehot, module vs fixture: %.2e (all), per question median %.2e max %.2e;  log-probs %.2e" % (
    l2rel(dl_mod, dl_ref), float(np.median([l2rel(dl_mod[i], dl_ref[i]) for i in range(B)])), max(l2rel(dl_mod[i], dl_ref[i]) for i in range(B)),
    gold.rel_err(lp.detach().cpu().numpy(), g["log_probs"])))

def per_q(a, ref):
    return np.array([l2rel(a[i], ref[i]) for i in range(B)])
for name, a, ref in (("dx module vs fixture", xt.grad.cpu().numpy(), g["dx"]), ("dq module vs fixture", qt.grad.cpu().numpy(), g["dq"]),
                     ("dx module vs fp32-with-kernel-gates", xt.grad.cpu().numpy(), dx1), ("dq module vs fp32-with-kernel-gates", qt.grad.cpu().numpy(), dq1),
                     ("dx fp32-with-kernel-gates vs fixture", dx1, g["dx"])):
    e = per_q(a, ref); o = np.argsort(-e)
    print("%-40s per question: median %.2e  p90 %.2e  worst five %s (questions %s)" % (name, np.median(e), np.quantile(e, 0.9), " ".join("%.1e" % e[i] for i in o[:5]), o[:5].tolist()))
for eps in (1e-4, 3e-4):
    dx2, dq2, _ = run(gates, eps)
    e = per_q(dq2, dq1)
    print("fp32 torch, kernel gates, x_g perturbed by %.0e relative (what a 16-bit forward does to f_phi's input): dx %.2e dq %.2e vs the unperturbed run; per question dq median %.2e worst %.2e" % (
        eps, l2rel(dx2, dx1), l2rel(dq2, dq1), np.median(e), e.max()))
