"""CPU experiment: weight-gradient error if the stored activations H_l (wgrad operand) were fp8 (e4m3, per-tensor scale) instead of
bf16.  Formula weights, relational layer at B=8, n=64 (M = 32768 pair rows); dZ stays bf16."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from oracle import formula, rn_oracle as O
torch.set_num_threads(8)
hyp = formula.HYP["original-fp"]; B, n, k, Q = 8, 64, 26, 128
sd = formula.formula_rl_state(hyp, 31)
x = torch.from_numpy(formula.formula_objects(B, n, k, 32)); q = torch.from_numpy(formula.hash_uniform((B, Q), 33, -1, 1)); lab = torch.from_numpy(formula.hash_ints((B,), 34, 0, 28))
rl = O.RelationalLayerOracle(hyp["rl_in_size"], 28, Q, hyp); rl.load_state_dict({k_: torch.from_numpy(v) for k_, v in sd.items()}); rl.eval()
acts, grads = {}, {}
for i, l in enumerate(rl.g_layers):
    l.register_forward_hook(lambda m, inp, out, i=i: acts.__setitem__(i, inp[0].detach()))
    l.register_full_backward_hook(lambda m, gi, go, i=i: grads.__setitem__(i, go[0].detach()))
lp = rl(x, q); torch.nn.functional.nll_loss(lp, lab).backward()
def l2(a, b): return float((a - b).norm() / b.norm())
for l in (1, 2, 3):
    Hin, dZ = acts[l].double(), grads[l].double()              # input of layer l (post-ReLU H_{l-1}), grad of its pre-activation
    ref = dZ.t() @ Hin
    dZb = grads[l].bfloat16().double()
    Hb = acts[l].bfloat16().double()
    s = 448.0 / float(acts[l].max())
    H8 = (acts[l] * s).to(torch.float8_e4m3fn).float().double() / s
    s5 = 57344.0 / float(acts[l].max())
    H5 = (acts[l] * s5).to(torch.float8_e5m2).float().double() / s5
    print("layer %d: dW rel-L2 error  bf16 dZ x bf16 H: %.2e   bf16 dZ x e4m3 H: %.2e   bf16 dZ x e5m2 H: %.2e   (fraction of H == 0: %.2f; e4m3 underflow to 0: %.3f)"
          % (l, l2(dZb.t() @ Hb, ref), l2(dZb.t() @ H8, ref), l2(dZb.t() @ H5, ref), float((acts[l] == 0).float().mean()), float(((H8 == 0) & (Hin != 0)).double().mean())))
print("--- dZ in fp8 as well (per-tensor scale by max |dZ|)")
for l in (1, 2, 3):
    Hin, dZ = acts[l].double(), grads[l].double()
    ref = dZ.t() @ Hin
    s = 448.0 / float(acts[l].max()); H8 = (acts[l] * s).to(torch.float8_e4m3fn).float().double() / s
    g = grads[l]; gm = float(g.abs().max())
    out = []
    for name, dt, mx in (("e4m3", torch.float8_e4m3fn, 448.0), ("e5m2", torch.float8_e5m2, 57344.0)):
        sz = mx / gm
        Z8 = (g * sz).to(dt).float().double() / sz
        out.append("%s dZ x e4m3 H: %.2e (dZ underflow to 0: %.3f)" % (name, l2(Z8.t() @ H8, ref), float(((Z8 == 0) & (dZ != 0)).double().mean())))
        # bias gradient = column sums of dZ
        out.append("db %s: %.2e" % (name, l2(Z8.sum(0), dZ.sum(0))))
    print("layer %d: " % l + "   ".join(out), "  | dZ dynamic range: max %.2e, median nonzero %.2e" % (gm, float(g[g != 0].abs().median())))
