"""CPU emulation: which g layers need the lo (second) weight pass for the 1e-3 log-prob bar?  Released checkpoints, B=4."""
import sys, os, itertools
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import gold
from oracle import formula, rn_oracle as O
torch.set_num_threads(8)

def run(tag):
    g = gold.load(tag); cfg = g["meta"]["cfg"]; hyp = formula.HYP[cfg]
    m = O.RNOracle(formula.QDICT, formula.ADICT, hyp)
    m.load_state_dict({k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("sd/")}, strict=False); m.eval()
    img = torch.from_numpy(formula.hash_uniform((4, 3, 128, 128), g["meta"]["img_seed"], 0.0, 1.0))
    qst = torch.from_numpy(formula.hash_ints((4, 20), g["meta"]["qst_seed"], 1, formula.QDICT + 1))
    with torch.no_grad():
        conv = m.conv(img); b, kk, d, _ = conv.shape
        coords = torch.from_numpy(O.coord_table(d))
        p = np.arange(d * d)
        x = torch.cat([conv.view(b, kk, d * d), coords[p % d][None, None, :].expand(b, 1, -1), coords[p // d][None, None, :].expand(b, 1, -1)], 1).permute(0, 2, 1).contiguous()
        q = m.text(qst)
        ref = torch.from_numpy(g["log_probs"])
        Ws = [l.weight.detach() for l in m.rl.g_layers]; bs = [l.bias.detach() for l in m.rl.g_layers]
        inj = hyp["question_injection_position"]; n = d * d; k = 26; Q = 128
        def f16(t): return t.half().float()
        def chain(lo_layers, act="f16", wfmt="f16"):
            rnd = (lambda t: t.half().float()) if act == "f16" else (lambda t: t.bfloat16().float())
            wr = (lambda t: t.half().float()) if wfmt == "f16" else (lambda t: t.bfloat16().float())
            def W_eff(l, W):
                hi = wr(W); 
                return hi + wr(W - hi) if l in lo_layers else hi
            outs = []
            for bi in range(b):
                xb = x[bi]; W0 = Ws[0]
                a = rnd(xb) @ W_eff(0, W0[:, :k]).t()                      # (n_j, 256)  x_j part on the MFMA
                v = xb @ W0[:, k:2 * k].t() + bs[0] + (q[bi] @ W0[:, 2 * k:].t() if inj == 0 else 0)   # fp32 bias row per i
                Hc = torch.relu(a[None, :, :] + v[:, None, :]).reshape(n * n, 256)      # rows (i, j)
                for l in range(1, 4):
                    W = Ws[l]
                    if l == inj:
                        z = rnd(Hc) @ W_eff(l, W[:, :256]).t() + (q[bi] @ W[:, 256:].t() + bs[l])
                    else:
                        z = rnd(Hc) @ W_eff(l, W).t() + bs[l]
                    Hc = torch.relu(z)
                outs.append(Hc.sum(0))
            xg = torch.stack(outs)
            f = m.rl
            h = torch.relu(f.f_fc1(xg)); h = torch.relu(f.f_fc2(h)); return torch.log_softmax(f.f_fc3(h), 1)
        print(tag)
        for lo in [(), (0, 1, 2, 3), (1, 2, 3), (2, 3), (3,), (1,), (2,), (1, 2), (1, 3), (0,)]:
            lp = chain(set(lo))
            print("   fp16 acts, lo pass on layers %-14s rel err %.2e" % (lo, float((lp - ref).abs().max() / ref.abs().max())))
        for lo in [(), (0, 1, 2, 3)]:
            lp = chain(set(lo), act="bf16", wfmt="bf16")
            print("   bf16 acts/weights, lo on %-14s rel err %.2e" % (lo, float((lp - ref).abs().max() / ref.abs().max())))
        lp = chain(set((0, 1, 2, 3)), act="bf16", wfmt="f16")
        print("   bf16 acts, fp16 hi+lo weights all layers: rel err %.2e" % float((lp - ref).abs().max() / ref.abs().max()))
run("pretrained_original_fp"); run("pretrained_ir_fp")
