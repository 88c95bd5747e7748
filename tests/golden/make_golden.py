"""Generate the golden vectors under tests/golden/ from the REFERENCE itself.

Run in the build container only (it imports /root/reference/model.py, which does
not exist on the GPU box):   python tests/golden/make_golden.py

Every fixture = closed-form inputs (oracle/formula.py seeds, regenerated at test
time) + outputs recorded from the reference's own PyTorch code on the CPU in fp32.
Fixture IDs follow SURVEY.md section 8c.  Only arrays are stored -- no reference code.
"""
import json
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

import model as refmodel  # noqa: E402  (the reference)
from oracle import formula  # noqa: E402

torch.set_num_threads(8)
REF_HYP = json.load(open("/root/reference/config.json"))["hyperparams"]
for name, h in formula.HYP.items():          # the restated table must equal the reference's
    assert REF_HYP[name] == h, name


class Args:
    qdict_size = formula.QDICT
    adict_size = formula.ADICT


def sample_idx(shape, seed, count=256):
    n = int(np.prod(shape))
    return formula.hash_ints((min(count, n),), seed, 0, n)


def build_ref_rl(cfg, seed, gain=1.0):
    hyp = REF_HYP[cfg]
    rl = refmodel.RelationalLayer(hyp["rl_in_size"], formula.ADICT, hyp["lstm_hidden"], hyp)
    sd = formula.formula_rl_state(hyp, seed, gain=gain)
    rl.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    return rl, hyp


def record_rl(tag, cfg, b, n, seed, **kw):
    """Retry with seed+1000 until no f_phi pre-activation sits within 1e-5 (relative) of
    its ReLU kink: a unit that close to zero flips between two fp32 summation orders and
    makes that sample's *gradient* a coin toss (the forward value is unaffected)."""
    while True:
        margin = _record_rl(tag, cfg, b, n, seed, **kw)
        if margin > 1e-5:
            return
        seed += 1000


def _record_rl(tag, cfg, b, n, seed, strided=False, train_dropout=False, full=False, full_grads=False, extra=None, light=False):
    rl, hyp = build_ref_rl(cfg, seed)
    k, Q = hyp["rl_in_size"] // 2, hyp["lstm_hidden"]
    x_np = formula.formula_objects(b, n, k, seed + 1, from_pixels=not hyp["state_description"])
    q_np = formula.hash_uniform((b, Q), seed + 2, -1.0, 1.0)
    lab = formula.hash_ints((b,), seed + 3, 0, formula.ADICT)
    if strided:   # physical layout (B,k,n) viewed as (B,n,k): what RN.forward hands to rl (model.py:200-201)
        x = torch.from_numpy(np.ascontiguousarray(x_np.transpose(0, 2, 1))).permute(0, 2, 1)
    else:
        x = torch.from_numpy(x_np.copy())
    x.requires_grad_(True)
    q = torch.from_numpy(q_np.copy()).requires_grad_(True)
    out = {}
    captured = {}
    if train_dropout:
        rl.train()
        torch.manual_seed(1234)
        def hook(_m, i, o):
            assert (i[0] != 0).all()
            captured["mask"] = (o / i[0]).detach().numpy().astype(np.float32)
        rl.dropout.register_forward_hook(hook)
    else:
        rl.eval()
    inj = hyp["question_injection_position"]
    if inj != 0:
        rl.g_layers[inj].register_forward_hook(lambda _m, i, o: captured.__setitem__("inj_in", i[0].detach().numpy()))
    last = len(hyp["g_layers"]) - 1
    rl.g_layers[last].register_forward_hook(lambda _m, i, o: captured.__setitem__("zL", o.detach()))
    pre = {}
    rl.f_fc1.register_forward_hook(lambda _m, i, o: pre.__setitem__("z1", o.detach()))
    rl.f_fc2.register_forward_hook(lambda _m, i, o: pre.__setitem__("z2", o.detach()))
    lp = rl(x, q)
    margin = min(float(pre[z].abs().min() / pre[z].abs().max()) for z in ("z1", "z2"))
    if margin <= 1e-5:
        return margin
    loss = F.nll_loss(lp, torch.from_numpy(lab))
    loss.backward()
    hL = torch.relu(captured["zL"])
    out["x_g"] = hL.view(b, n * n, -1).sum(1).numpy()
    out["log_probs"] = lp.detach().numpy()
    out["loss"] = np.array(loss.item(), dtype=np.float32)
    if light:   # big-shape record kept KB-sized (SURVEY 8c "norms at B=32"): norm + sampled entries instead of the full dx
        dxn = x.grad.numpy()
        idx = sample_idx(dxn.shape, seed + 23, 4096)
        out["dx_norm"] = np.array(np.linalg.norm(dxn.astype(np.float64)), dtype=np.float64)
        out["dx_sample_idx"] = idx
        out["dx_sample"] = dxn.reshape(-1)[idx].copy()
    else:
        out["dx"] = x.grad.numpy().copy()
    out["dq"] = q.grad.numpy().copy()
    for name, p in rl.named_parameters():
        gnp = p.grad.numpy()
        if name.endswith("bias") or (full_grads and gnp.size <= 70000) or gnp.size <= 8192:
            out["grad/" + name] = gnp.copy()
        else:
            idx = sample_idx(gnp.shape, seed + 17, 1024)
            out["gradnorm/" + name] = np.array(np.linalg.norm(gnp.astype(np.float64)), dtype=np.float64)
            out["gradsample_idx/" + name] = idx
            out["gradsample/" + name] = gnp.reshape(-1)[idx].copy()
    if train_dropout:
        out["dropout_mask"] = captured["mask"]
    if inj != 0:
        rows = sample_idx((b * n * n,), seed + 19, 64)
        out["inj_rows"] = rows
        out["inj_in_rows"] = captured["inj_in"][rows].copy()
    if full:   # small case: keep the first pair rows too (pins row order / column order)
        P = torch.cat([x.detach()[:, None].expand(b, n, n, k), x.detach()[:, :, None].expand(b, n, n, k)], 3)
        out["pair_rows_b0"] = P[0].reshape(n * n, 2 * k)[: 4 * n].numpy().copy()
    out["meta"] = np.array(json.dumps(dict(cfg=cfg, b=b, n=n, seed=seed, strided=strided,
                                           train_dropout=train_dropout, f_margin=margin)))
    if extra:
        out.update(extra)
    np.savez_compressed(os.path.join(HERE, tag + ".npz"), **out)
    print(tag, {k_: (v.shape if hasattr(v, "shape") else v) for k_, v in out.items() if not k_.startswith("grad")},
          "loss", loss.item(), "margin", margin)
    return margin


def record_e2e(tag, cfg, b, seed, img_hw=128, T=20):
    hyp = REF_HYP[cfg]
    m = refmodel.RN(Args, hyp)
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    sd = formula.formula_fill_state(shapes, seed)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    m.eval()
    img = formula.hash_uniform((b, 3, img_hw, img_hw), seed + 1, 0.0, 1.0)
    qst = formula.hash_ints((b, T), seed + 2, 1, formula.QDICT + 1)
    with torch.no_grad():
        lp = m(torch.from_numpy(img), torch.from_numpy(qst))
        conv = m.conv(torch.from_numpy(img))
        qemb = m.text(torch.from_numpy(qst))
    np.savez_compressed(os.path.join(HERE, tag + ".npz"), log_probs=lp.numpy(), conv_out=conv.numpy(),
                        qst_emb=qemb.numpy(), state_names=np.array(json.dumps({k: list(v) for k, v in shapes.items()})),
                        meta=np.array(json.dumps(dict(cfg=cfg, b=b, seed=seed, img_hw=img_hw, T=T))))
    print(tag, lp.shape, conv.shape)


def record_extract(tag, cfg, b, seed, layer_idx=2, q_seed=None):
    """extract.py:49-74 semantics: hook the *input* of g_layers[k], strip the
    trailing question columns, L2-normalise per pair, max / mean over pairs.
    q_seed: a non-zero question embedding (pins how the question enters the layers in front of the hook)."""
    hyp = REF_HYP[cfg]
    rl = refmodel.RelationalLayer(hyp["rl_in_size"], formula.ADICT, hyp["lstm_hidden"], hyp, extraction=True)
    sd = formula.formula_rl_state(hyp, seed)
    rl.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    rl.eval()
    n, k, Q = 64, hyp["rl_in_size"] // 2, hyp["lstm_hidden"]
    x = torch.from_numpy(formula.formula_objects(b, n, k, seed + 1))
    q = torch.zeros(b, Q)                       # extract.py:103 feeds an all-zero question
    if q_seed is not None:
        q = torch.from_numpy(formula.hash_uniform((b, Q), q_seed, -1.0, 1.0))
    got = {}
    def hook(_m, i, o):
        feats = i[0].detach().view(b, n * n, -1)
        if layer_idx == hyp["question_injection_position"]:
            feats = feats[:, :, :-Q]
        feats = feats / feats.norm(2, 2, keepdim=True).clamp_min(1e-12)
        got["max"] = feats.max(1)[0].numpy()
        got["avg"] = feats.mean(1).numpy()
    rl.g_layers[layer_idx].register_forward_hook(hook)
    with torch.no_grad():
        assert rl(x, q) is None                 # model.py:147-148
    np.savez_compressed(os.path.join(HERE, tag + ".npz"), max=got["max"], avg=got["avg"],
                        meta=np.array(json.dumps(dict(cfg=cfg, b=b, seed=seed, layer_idx=layer_idx, q_seed=q_seed))))
    print(tag, got["max"].shape)


def record_pretrained():
    """The two released checkpoints as plain arrays (MIT-licensed data, README.md:86-95),
    keys with the DataParallel 'module.' prefix stripped (train.py:271-274)."""
    for fn, tag in (("original_fp_epoch_493.pth", "pretrained_original_fp"), ("ir_fp_epoch_312.pth", "pretrained_ir_fp")):
        sd = torch.load("/root/reference/pretrained_models/" + fn, weights_only=False, map_location="cpu")
        arrs = {k.replace("module.", "", 1): v.numpy() for k, v in sd.items()}
        cfg = "original-fp" if "original" in fn else "ir-fp"
        m = refmodel.RN(Args, REF_HYP[cfg])
        m.load_state_dict({k: torch.from_numpy(v) for k, v in arrs.items()}, strict=False)
        m.eval()
        img = formula.hash_uniform((4, 3, 128, 128), 77, 0.0, 1.0)
        qst = formula.hash_ints((4, 20), 78, 1, formula.QDICT + 1)
        with torch.no_grad():
            lp = m(torch.from_numpy(img), torch.from_numpy(qst)).numpy()
        np.savez_compressed(os.path.join(HERE, tag + ".npz"), **{"sd/" + k: v for k, v in arrs.items()},
                            log_probs=lp, meta=np.array(json.dumps(dict(cfg=cfg, img_seed=77, qst_seed=78))))
        print(tag, len(arrs), lp.shape)


def record_train_traj(tag, cfg, b, steps, seed, lr=1e-4):
    """N1: the reference's training step (train.py:36-48) -- zero_grad, forward, nll_loss, backward,
    clip_grad_norm(50), Adam(weight_decay=1e-4) -- run for a few steps on fixed closed-form batches with
    the reference's own model (dropout overridden to 0 as train.py:207-208 allows, train-mode BatchNorm).
    Questions are reversed and labels shifted to 0-based exactly as utils.load_tensor_data does
    (utils.py:138-149; restated here because utils.py's Variable(volatile=...) idiom is torch-0.3 only)."""
    hyp = dict(REF_HYP[cfg], dropout=0.0)
    m = refmodel.RN(Args, hyp)
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    sd = formula.formula_fill_state(shapes, seed)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    m.train()
    opt = torch.optim.Adam(m.parameters(), lr=lr, weight_decay=1e-4)
    losses, norms = [], []
    for s in range(steps):
        img = torch.from_numpy(formula.hash_uniform((b, 3, 128, 128), seed + 10 * s + 1, 0.0, 1.0))
        qst = torch.from_numpy(formula.hash_ints((b, 20), seed + 10 * s + 2, 1, formula.QDICT + 1))
        ans = torch.from_numpy(formula.hash_ints((b, 1), seed + 10 * s + 3, 1, formula.ADICT + 1))      # 1-based like the dataset
        qst = qst.index_select(1, torch.arange(qst.size(1) - 1, -1, -1).long())
        label = (ans - 1).squeeze(1)
        opt.zero_grad()
        loss = F.nll_loss(m(img, qst), label)
        loss.backward()
        norms.append(float(torch.nn.utils.clip_grad_norm_(m.parameters(), 50.0)))
        opt.step()
        losses.append(loss.item())
    final = {k: v.detach().numpy() for k, v in m.state_dict().items() if k in ("rl.g_layers.0.bias", "rl.f_fc3.bias", "conv.batchNorm4.running_mean")}
    np.savez_compressed(os.path.join(HERE, tag + ".npz"), losses=np.array(losses), grad_norms=np.array(norms),
                        state_names=np.array(json.dumps({k: list(v) for k, v in shapes.items()})),
                        **{"final/" + k: v for k, v in final.items()},
                        meta=np.array(json.dumps(dict(cfg=cfg, b=b, steps=steps, seed=seed, lr=lr))))
    print(tag, "losses", losses, "norms", norms)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "traj":
        record_train_traj("G-traj", "original-fp", 8, 4, seed=91)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "extract":
        for cfg_ in ("original-fp", "ir-fp"):
            for li in range(4):
                record_extract("G-extract-%s-%d" % (cfg_, li), cfg_, 3, seed=82 + li, layer_idx=li, q_seed=90 + li)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "stress":
        # BASELINE.json configs[4] at its real size (B=32, 14x14 grid: M = 1,229,312 pair rows; ~12 GB, ~1 min on 8 cores)
        record_rl("G-fp196-b32", "original-fp", 32, 196, seed=52, light=True)
        sys.exit(0)
    record_rl("G-sd4", "original-sd", 4, 12, seed=11, full=True)
    record_rl("G-irsd4", "ir-sd", 4, 12, seed=12, full=True)
    record_rl("G-fp-small", "original-fp", 2, 64, seed=21, strided=True, full=True, full_grads=True)
    record_rl("G-ir-small", "ir-fp", 2, 64, seed=22, strided=True, full=True, full_grads=True)
    record_rl("G-fp64", "original-fp", 64, 64, seed=31)
    record_rl("G-ir64", "ir-fp", 64, 64, seed=41)
    record_rl("G-fp196", "original-fp", 2, 196, seed=51)
    record_rl("G-fp196-b32", "original-fp", 32, 196, seed=52, light=True)
    record_rl("G-drop", "original-fp", 4, 64, seed=61, train_dropout=True, full=True)
    record_e2e("G-e2e", "original-fp", 4, seed=71)
    record_e2e("G-e2e-ir", "ir-fp", 4, seed=72)
    record_extract("G-extract", "ir-fp", 4, seed=81)
    for cfg_ in ("original-fp", "ir-fp"):       # every hook position of both 256-wide models, with a real question (N3)
        for li in range(4):
            record_extract("G-extract-%s-%d" % (cfg_, li), cfg_, 3, seed=82 + li, layer_idx=li, q_seed=90 + li)
    record_pretrained()
    record_train_traj("G-traj", "original-fp", 8, 4, seed=91)
