#!/usr/bin/env python3
"""Where the gradient error of a 16-bit forward comes from (CPU only, torch fp32): the relational layer of fixture G-fp64
(B = 64, n = 64) is differentiated in fp32 arithmetic THREE times --
  (a) as the reference does it,
  (b) with the ReLU gates of a forward pass whose operands are rounded to fp16 (fp32 accumulate: the f16s arithmetic, weights un-dithered),
  (c) the same with bf16 operands,
everything else (values, sums, the whole backward) in fp32.  (b) and (c) differ from (a) ONLY in the gates that flip where a
pre-activation is within the forward's rounding error of zero.  Prints the flipped fraction per layer and the relative L2 error of
dx / dq / the bias gradients against the fixture -- to be read beside the parity block of a bench line (dx 1.39e-2 in f16s mode).
Then the backward chain's own arithmetic on top of (b): dZ_l and W_l^T rounded to bf16 (what the kernels do) or to fp16 in every
dgrad step (fp32 accumulate, layer 0's reductions and dx / dq in fp32)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import gold

g = gold.load("G-fp64")
hyp, sd, x, q, lab = gold.rl_case(g["meta"])
B, n, k = x.shape
Q = q.shape[1]
W = [torch.from_numpy(sd["g_layers.%d.weight" % l]) for l in range(4)]
b = [torch.from_numpy(sd["g_layers.%d.bias" % l]) for l in range(4)]
fW = [torch.from_numpy(sd["f_fc%d.weight" % (i + 1)]) for i in range(3)]
fb = [torch.from_numpy(sd["f_fc%d.bias" % (i + 1)]) for i in range(3)]
labt = torch.from_numpy(lab)


def l2rel(a, ref):
    a = np.asarray(a, np.float64); ref = np.asarray(ref, np.float64)
    return float(np.linalg.norm(a - ref) / np.linalg.norm(ref))


def pairs(xt, qt):
    xj = xt[:, None, :, :].expand(B, n, n, k)                      # model.py:117-127: [x_j | x_i | q], row (b, i, j)
    xi = xt[:, :, None, :].expand(B, n, n, k)
    qq = qt[:, None, None, :].expand(B, n, n, Q)
    return torch.cat([xj, xi, qq], 3).reshape(B * n * n, 2 * k + Q)


def gates_of(dtype):
    """gates of a forward whose operands (activations, weights) are rounded to `dtype`, fp32 accumulate"""
    with torch.no_grad():
        h = pairs(torch.from_numpy(x), torch.from_numpy(q))
        out = []
        for l in range(4):
            if dtype is None:
                z = h @ W[l].t() + b[l]
            else:
                z = h.to(dtype).float() @ W[l].to(dtype).float().t() + b[l]
            out.append(z > 0)
            h = torch.relu(z)
    return out


def grads(gates):
    xt = torch.from_numpy(x).clone().requires_grad_(True); qt = torch.from_numpy(q).clone().requires_grad_(True)
    bs = [t.clone().requires_grad_(True) for t in b]
    h = pairs(xt, qt)
    for l in range(4):
        h = (h @ W[l].t() + bs[l]) * gates[l]                      # fp32 values, the given gates
    xg = h.view(B, n * n, -1).sum(1)
    f = torch.relu(xg @ fW[0].t() + fb[0])
    f = torch.relu(f @ fW[1].t() + fb[1])                          # (eval: no dropout, like the fixture)
    lp = torch.log_softmax(f @ fW[2].t() + fb[2], 1)
    torch.nn.functional.nll_loss(lp, labt).backward()
    return xt.grad.numpy(), qt.grad.numpy(), [t.grad.numpy() for t in bs], lp.detach().numpy()


ref_gates = gates_of(None)
for name, dt in (("fp32 gates (the reference's own)", None), ("gates of an fp16-operand forward", torch.float16), ("gates of a bf16-operand forward", torch.bfloat16)):
    gs = gates_of(dt)
    flips = [float((a != r).float().mean()) for a, r in zip(gs, ref_gates)]
    dx, dq, dbs, lp = grads(gs)
    print("%-36s flipped gates per layer %s   dx %.2e  dq %.2e  bias grads (max) %.2e   log-probs %.1e" % (
        name, " ".join("%.1e" % f for f in flips), l2rel(dx, g["dx"]), l2rel(dq, g["dq"]),
        max(l2rel(dbs[l], g["grad/g_layers.%d.bias" % l]) for l in range(4)), gold.rel_err(lp, g["log_probs"])))


def chain_backward(gates, dt):
    """fp32 forward with the given gates; g_theta backward with dZ / W^T rounded to dt at every step (None: fp32)"""
    rnd = (lambda t: t) if dt is None else (lambda t: t.to(dt).float())
    with torch.no_grad():
        h = pairs(torch.from_numpy(x), torch.from_numpy(q))
        for l in range(4):
            h = (h @ W[l].t() + b[l]) * gates[l]
    xg = h.view(B, n * n, -1).sum(1).requires_grad_(True)
    f = torch.relu(xg @ fW[0].t() + fb[0]); f = torch.relu(f @ fW[1].t() + fb[1])
    torch.nn.functional.nll_loss(torch.log_softmax(f @ fW[2].t() + fb[2], 1), labt).backward()
    with torch.no_grad():
        dxg = xg.grad
        dz = rnd(dxg.repeat_interleave(n * n, 0) * gates[3])       # (the kernels keep dZ_3 un-rounded for the wgrad; the chain rounds its operand)
        for l in (3, 2, 1):
            dz = (dz @ rnd(W[l])) * gates[l - 1]
            if l > 1:
                dz = rnd(dz)                                       # dZ_0 itself is reduced un-rounded (on-chip pair reductions)
        dz = dz.view(B, n, n, -1)
        Rj, Ri, Rq = dz.sum(1), dz.sum(2), dz.sum((1, 2))          # over i / over j / both
        dx = Rj @ W[0][:, :k] + Ri @ W[0][:, k:2 * k]
        dq = Rq @ W[0][:, 2 * k:]
    return dx.numpy(), dq.numpy()


g16 = gates_of(torch.float16)
for name, dt in (("fp32 chain", None), ("bf16 chain (the kernels')", torch.bfloat16), ("fp16 chain", torch.float16)):
    dx, dq = chain_backward(g16, dt)
    print("fp16-forward gates + %-26s dx %.2e  dq %.2e" % (name, l2rel(dx, g["dx"]), l2rel(dq, g["dq"])))
