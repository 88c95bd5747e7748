"""CPU emulation of the fp8 wgrad operands exactly as the kernels would store them: e4m3 with one power-of-two scale per
(32 pair rows x 32 features) block (amax / scale in [128, 256)), for H_0..H_2 and dZ_1, dZ_2; the last layer's dZ_3 is never
quantised (mask^T H_2 per question, times dxg in fp32).  Prints the relative L2 error of dW_l / db_l against fp64."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from oracle import formula, rn_oracle as O
torch.set_num_threads(16)
def l2(a, b): return float((a - b).norm() / b.norm())
def q8(t, static=None):
    M, F = t.shape
    blk = t.reshape(M // 32, 32, F // 32, 32)
    if static is None:
        amax = blk.abs().amax(dim=(1, 3), keepdim=True).clamp_min(1e-30)
        sc = torch.exp2(torch.floor(torch.log2(amax)) - 7)
    else:
        sc = torch.full((1, 1, 1, 1), static)
    qv = (blk / sc).to(torch.float8_e4m3fn).float() * sc
    return qv.reshape(M, F)
def run(name, hyp, sd, x, q, lab):
    B, n = x.shape[0], x.shape[1]
    rl = O.RelationalLayerOracle(hyp["rl_in_size"], 28, q.shape[1], hyp); rl.load_state_dict(sd); rl.eval()
    acts, grads = {}, {}
    for i, l in enumerate(rl.g_layers):
        l.register_forward_hook(lambda m, inp, out, i=i: acts.__setitem__(i, inp[0].detach()))
        l.register_full_backward_hook(lambda m, gi, go, i=i: grads.__setitem__(i, go[0].detach()))
    lp = rl(x, q); torch.nn.functional.nll_loss(lp, lab).backward()
    print("== %s: B=%d n=%d" % (name, B, n))
    for l in (1, 2, 3):
        H, dZ = acts[l][:, :256], grads[l]
        ref = dZ.double().t() @ H.double()
        now = dZ.bfloat16().double().t() @ H.bfloat16().double()
        H8 = q8(H); Hs1 = q8(H, 1.0); Hs4 = q8(H, 0.25)
        Z = dZ if l == 3 else q8(dZ)
        Zb = dZ.bfloat16().float()
        print(" layer %d  H fp8 (block scale) x dZ bf16: dW err %.2e; H e5m2-ish n/a; H fp8 static1 x dZ bf16: %.2e" % (l, l2(Zb.double().t() @ H8.double(), ref), l2(Zb.double().t() @ Hs1.double(), ref)))
        print(" layer %d  |H| max %.3g  median>0 %.3g | dW err: bf16xbf16 %.2e | blockscale fp8 %.2e | static H scale 1: %.2e, H/4: %.2e | db err fp8 %.2e (bf16 %.2e)"
              % (l, float(H.max()), float(H[H > 0].median()), l2(now, ref), l2(Z.double().t() @ H8.double(), ref),
                 l2(Z.double().t() @ Hs1.double(), ref), l2(Z.double().t() @ Hs4.double(), ref),
                 l2(Z.double().sum(0), dZ.double().sum(0)), l2(dZ.bfloat16().double().sum(0), dZ.double().sum(0))))
hyp = formula.HYP["original-fp"]; B, n, k, Q = 8, 64, 26, 128
sd = {k_: torch.from_numpy(v) for k_, v in formula.formula_rl_state(hyp, 31).items()}
x = torch.from_numpy(formula.formula_objects(B, n, k, 32)); q = torch.from_numpy(formula.hash_uniform((B, Q), 33, -1, 1)); lab = torch.from_numpy(formula.hash_ints((B,), 34, 0, 28))
run("formula weights", hyp, sd, x, q, lab)
for tag, cfg in (("pretrained_original_fp", "original-fp"),):
    ck = np.load(os.path.join(ROOT, "tests", "golden", tag + ".npz"))
    import json
    meta = json.loads(str(ck["meta"])) if str(ck["meta"]).startswith("{") else eval(str(ck["meta"]))
    hyp = formula.HYP[cfg]
    m = O.RNOracle(formula.QDICT, formula.ADICT, hyp)
    m.load_state_dict({k_[3:]: torch.from_numpy(ck[k_]) for k_ in ck.files if k_.startswith("sd/")}, strict=False); m.eval()
    Bc = 8
    img = torch.from_numpy(formula.hash_uniform((Bc, 3, 128, 128), meta["img_seed"], 0.0, 1.0)); qst = torch.from_numpy(formula.hash_ints((Bc, 20), meta["qst_seed"], 1, formula.QDICT + 1))
    with torch.no_grad():
        xo, qe = m.objects(img), m.text(qst)
    sdr = {k_[6:]: torch.from_numpy(ck[k_]) for k_ in ck.files if k_.startswith("sd/rl.")}
    run(tag + " (hash images)", hyp, sdr, xo, qe, torch.from_numpy(formula.hash_ints((Bc,), 5, 0, 28)))
