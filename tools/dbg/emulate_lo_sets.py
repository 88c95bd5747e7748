#!/usr/bin/env python3
"""CPU emulation of the f16s forward arithmetic (fp16 operand registers x fp16 hi [+ lo] weight fragments, fp32 accumulate; factored
first layer: fp16 object rows x W0a, the x_i / q / bias bracket exact) on the RELEASED checkpoints: which layers need the second
(lo) pass?  Prints the worst max-norm relative log-prob error over the questions for every subset of layers that keeps it.
usage: emulate_lo_sets.py [original_fp | ir_fp] [questions]"""
import itertools, os, sys
import numpy as np, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
from oracle import formula, rn_oracle as O

name = sys.argv[1] if len(sys.argv) > 1 else "original_fp"
NQ = int(sys.argv[2]) if len(sys.argv) > 2 else 16
g = np.load(os.path.join(ROOT, "tests", "golden", "pretrained_%s.npz" % name))
import json
meta = json.loads(str(g["meta"]))
hyp = formula.HYP[meta["cfg"]]
m = O.RNOracle(formula.QDICT, formula.ADICT, hyp)
m.load_state_dict({k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd/")}, strict=False)
m.eval()
img = torch.from_numpy(formula.hash_uniform((NQ, 3, 128, 128), meta["img_seed"], 0.0, 1.0))
qst = torch.from_numpy(formula.hash_ints((NQ, 20), meta["qst_seed"], 1, formula.QDICT + 1))
with torch.no_grad():
    x = m.objects(img).numpy().astype(np.float64)            # (B, n, k)
    q = m.text(qst).numpy().astype(np.float64)
    ref = m(img, qst).numpy().astype(np.float64)
rl = m.rl
Ws = [l.weight.detach().numpy().astype(np.float32) for l in rl.g_layers]
bs = [l.bias.detach().numpy().astype(np.float64) for l in rl.g_layers]
fW = [getattr(rl, "f_fc%d" % i).weight.detach().numpy().astype(np.float64) for i in (1, 2, 3)]
fb = [getattr(rl, "f_fc%d" % i).bias.detach().numpy().astype(np.float64) for i in (1, 2, 3)]
inj = hyp["question_injection_position"]
B, n, k = x.shape
f16 = lambda a: np.asarray(a, np.float32).astype(np.float16).astype(np.float64)


def split(W, lo):
    hi = W.astype(np.float16).astype(np.float32)
    w = hi.astype(np.float64)
    if lo:
        w = w + (W - hi).astype(np.float16).astype(np.float64)
    return w


def run(lo_set):
    out = np.zeros((B, formula.ADICT))
    for b in range(B):
        W0 = Ws[0].astype(np.float64)
        xj = f16(x[b]) @ split(Ws[0][:, :k], 0 in lo_set).T                               # (n, 256): the MFMA part of layer 0
        br = x[b] @ W0[:, k:2 * k].T + bs[0] + (q[b] @ W0[:, 2 * k:].T if inj == 0 else 0.0)      # exact bracket per i
        h = np.maximum(xj[None, :, :] + br[:, None, :], 0).reshape(n * n, -1)
        for l in range(1, 4):
            a = f16(np.minimum(h, 65504.0))
            Wl = Ws[l]
            z = a @ split(Wl[:, :256], l in lo_set).T + bs[l]
            if l == inj:
                z = z + q[b] @ Wl[:, 256:].astype(np.float64).T
            h = np.maximum(z, 0)
        xg = h.sum(0)
        f1 = np.maximum(fW[0] @ xg + fb[0], 0)
        f2 = np.maximum(fW[1] @ f1 + fb[1], 0)
        z = fW[2] @ f2 + fb[2]
        out[b] = z - z.max() - np.log(np.exp(z - z.max()).sum())
    return float(np.abs(out - ref).max() / np.abs(ref).max())


print("config %s, %d questions; worst log-prob error (max-norm relative; bar 1e-3)" % (meta["cfg"], NQ))
for r in (range(5) if os.environ.get("ALL_SETS", "0") == "1" else ()):
    for s in itertools.combinations(range(4), r):
        print("lo pass on layers %-12s : %.2e" % (list(s), run(set(s))), flush=True)


# ---- dithered hi images: layer `l` hi-only, but tile t (256 pair rows) multiplies image t % V whose weights are RNE(W + d_v ulp(W)),
# d_v = (v + 0.5) / V - 0.5: the mean over the V images is within ulp / (2 V) of W, and the pair sum averages over the tiles
def ulp16(W):
    e = np.floor(np.log2(np.maximum(np.abs(W.astype(np.float64)), 2.0 ** -14)))
    return 2.0 ** (e - 10)


def run_dither(lo_set, dith, V):
    out = np.zeros((B, formula.ADICT))
    imgs = {l: [f16(Ws[l][:, :256].astype(np.float64) + ((v + 0.5) / V - 0.5) * ulp16(Ws[l][:, :256])) for v in range(V)] for l in dith}
    T = n * n // 256
    for b in range(B):
        W0 = Ws[0].astype(np.float64)
        xj = f16(x[b]) @ split(Ws[0][:, :k], 0 in lo_set).T
        br = x[b] @ W0[:, k:2 * k].T + bs[0] + (q[b] @ W0[:, 2 * k:].T if inj == 0 else 0.0)
        h = np.maximum(xj[None, :, :] + br[:, None, :], 0).reshape(n * n, -1)
        for l in range(1, 4):
            a = f16(np.minimum(h, 65504.0))
            if l in dith:
                z = np.empty((n * n, 256))
                for t in range(T):
                    z[256 * t:256 * (t + 1)] = a[256 * t:256 * (t + 1)] @ imgs[l][t % V].T
                z += bs[l]
            else:
                z = a @ split(Ws[l][:, :256], l in lo_set).T + bs[l]
            if l == inj:
                z = z + q[b] @ Ws[l][:, 256:].astype(np.float64).T
            h = np.maximum(z, 0)
        xg = h.sum(0)
        f1 = np.maximum(fW[0] @ xg + fb[0], 0)
        f2 = np.maximum(fW[1] @ f1 + fb[1], 0)
        z = fW[2] @ f2 + fb[2]
        out[b] = z - z.max() - np.log(np.exp(z - z.max()).sum())
    return float(np.abs(out - ref).max() / np.abs(ref).max())


for lo_set, dith, V in (((0,), (1, 2, 3), 4), ((0,), (1, 2, 3), 8), ((0, 2), (1, 3), 4), ((0,), (1, 2), 4), ((0,), (), 1)):
    print("lo on %-8s dithered hi images (%d) on %-10s: %.2e" % (list(lo_set), V, list(dith), run_dither(set(lo_set), dith, V)), flush=True)
