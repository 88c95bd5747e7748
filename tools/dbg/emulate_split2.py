"""CPU emulation over several input seeds: max log-prob error per choice of split layers (released checkpoints, fp16 activations)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import gold
from oracle import formula, rn_oracle as O
torch.set_num_threads(8)
CONFIGS = [(0, 1, 2, 3), (0, 1, 2), (0, 1), (0, 2)]
def run(tag, seeds):
    g = gold.load(tag); cfg = g["meta"]["cfg"]; hyp = formula.HYP[cfg]
    m = O.RNOracle(formula.QDICT, formula.ADICT, hyp)
    m.load_state_dict({k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("sd/")}, strict=False); m.eval()
    worst = {c: 0.0 for c in CONFIGS}; flips = {c: 0 for c in CONFIGS}
    for sd in seeds:
        img = torch.from_numpy(formula.hash_uniform((4, 3, 128, 128), sd, 0.0, 1.0))
        qst = torch.from_numpy(formula.hash_ints((4, 20), sd + 1, 1, formula.QDICT + 1))
        with torch.no_grad():
            ref = m(img, qst)
            conv = m.conv(img); b, kk, d, _ = conv.shape
            coords = torch.from_numpy(O.coord_table(d)); p = np.arange(d * d)
            x = torch.cat([conv.view(b, kk, d * d), coords[p % d][None, None, :].expand(b, 1, -1), coords[p // d][None, None, :].expand(b, 1, -1)], 1).permute(0, 2, 1).contiguous()
            q = m.text(qst)
            Ws = [l.weight.detach() for l in m.rl.g_layers]; bs = [l.bias.detach() for l in m.rl.g_layers]
            inj = hyp["question_injection_position"]; n = d * d; k = 26
            rnd = lambda t: t.half().float()
            for lo in CONFIGS:
                W_eff = lambda l, W: (rnd(W) + rnd(W - rnd(W))) if l in lo else rnd(W)
                outs = []
                for bi in range(b):
                    xb = x[bi]; W0 = Ws[0]
                    a = rnd(xb) @ W_eff(0, W0[:, :k]).t()
                    v = xb @ W0[:, k:2 * k].t() + bs[0] + (q[bi] @ W0[:, 2 * k:].t() if inj == 0 else 0)
                    Hc = torch.relu(a[None, :, :] + v[:, None, :]).reshape(n * n, 256)
                    for l in range(1, 4):
                        W = Ws[l]
                        z = rnd(Hc) @ W_eff(l, W[:, :256]).t() + ((q[bi] @ W[:, 256:].t() + bs[l]) if l == inj else bs[l])
                        Hc = torch.relu(z)
                    outs.append(Hc.sum(0))
                f = m.rl; h = torch.relu(f.f_fc1(torch.stack(outs))); h = torch.relu(f.f_fc2(h)); lp = torch.log_softmax(f.f_fc3(h), 1)
                worst[lo] = max(worst[lo], float((lp - ref).abs().max() / ref.abs().max())); flips[lo] += int((lp.argmax(1) != ref.argmax(1)).sum())
    print(tag, "over", len(seeds), "x B=4:")
    for lo in CONFIGS:
        print("   lo on %-14s worst rel err %.2e  argmax flips %d" % (lo, worst[lo], flips[lo]))
run("pretrained_original_fp", [77, 101, 202, 303, 404, 505]); run("pretrained_ir_fp", [77, 101, 202, 303, 404, 505])
