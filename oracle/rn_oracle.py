"""CPU oracle for the Relation-Network hot path  --  TEST INFRASTRUCTURE ONLY.

This file is the checker, never the product: only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it.  The shipped path (``relationnetworks-clevr_amd/``) never imports
anything from ``oracle/`` and fails loudly when the HIP extension is missing.

It restates, in fp32 on the CPU, the algorithm of the reference's hot path
(``/root/reference/model.py:60-223``):

* ``pair_matrix``            <- model.py:108-127 (+ :135-140 question concat)
* ``g_forward`` / ``rl_forward_np``  <- model.py:130-162
* ``rl_backward_np``         <- what autograd derives for the above (train.py:41-42)
* ``coord_table`` / ``grid_to_objects`` <- model.py:191-201, :208-218
* ``RelationalLayerOracle`` / ``RNOracle`` (torch, un-fused op sequence) <- model.py:9-223

Parity status: PINNED.  The reference has no tests or golden vectors of its own
(SURVEY.md section 4), so the pin is the reference itself: ``tests/golden/make_golden.py``
imports ``/root/reference/model.py`` in the build container, runs it on the
closed-form inputs of ``oracle/formula.py`` and commits inputs' seeds + outputs
as ``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks this file
against those vectors (fp32, <=1e-5 relative).
"""
from __future__ import annotations

import numpy as np

try:  # torch is only needed for the nn.Module flavoured oracle
    import torch
    import torch.nn as nn
    import torch.nn.functional as F
except Exception:  # pragma: no cover
    torch = None

F32 = np.float32


# --------------------------------------------------------------------------
# numpy restatement (explicit forward + hand-derived backward)
# --------------------------------------------------------------------------
def coord_table(d: int) -> np.ndarray:
    """model.py:209 -- ``torch.linspace(-d/2., d/2., d)`` in fp32.

    torch.linspace computes start + i*step for the lower half and
    end - (steps-1-i)*step for the upper half, in fp32 with a fused multiply-add;
    restated here so the table is bit-identical (checked against torch in the
    golden test)."""
    start, end = F32(-d / 2.0), F32(d / 2.0)
    if d == 1:
        return np.array([start], dtype=F32)
    step = F32((end - start) / F32(d - 1))
    out = np.empty(d, dtype=F32)
    half = d // 2
    for i in range(d):
        # torch's CPU kernel contracts the multiply-add (fma): one rounding of the exact
        # start + step*i  /  end - step*(d-1-i); float64 holds the fp32 product exactly.
        if i < half:
            out[i] = F32(np.float64(start) + np.float64(step) * i)
        else:
            out[i] = F32(np.float64(end) - np.float64(step) * (d - 1 - i))
    return out


def grid_to_objects(conv_out: np.ndarray) -> np.ndarray:
    """model.py:191-201.  (B,C,d,d) conv features -> (B, d*d, C+2) objects.

    object p = row*d + col;  channel C = x-coordinate = lin[col],
    channel C+1 = y-coordinate = lin[row]  (model.py:210-212)."""
    b, c, d, _ = conv_out.shape
    lin = coord_table(d)
    x = conv_out.reshape(b, c, d * d)
    cx = np.tile(lin[None, :], (d, 1)).reshape(d * d)      # coords[col]
    cy = np.tile(lin[:, None], (1, d)).reshape(d * d)      # coords[row]
    coords = np.stack([cx, cy], 0)[None].repeat(b, 0).astype(F32)
    x = np.concatenate([x, coords], 1)                      # (B, C+2, d*d)
    return np.ascontiguousarray(x.transpose(0, 2, 1))       # (B, d*d, C+2)


def pair_matrix(x: np.ndarray, q: np.ndarray | None) -> np.ndarray:
    """model.py:117-127 (+ :135-140 when the question is injected at layer 0).

    Row r = (b*n + i)*n + j  holds  [ x[b,j,:] | x[b,i,:] | q[b,:] ]."""
    b, n, k = x.shape
    x_i = np.broadcast_to(x[:, None, :, :], (b, n, n, k))   # [b,i,j] = x[b,j]
    x_j = np.broadcast_to(x[:, :, None, :], (b, n, n, k))   # [b,i,j] = x[b,i]
    parts = [x_i, x_j]
    if q is not None:
        parts.append(np.broadcast_to(q[:, None, None, :], (b, n, n, q.shape[1])))
    return np.concatenate(parts, 3).reshape(b * n * n, -1).astype(F32)


def append_question(h: np.ndarray, q: np.ndarray, b: int, n: int) -> np.ndarray:
    """model.py:135-140 -- question appended as the trailing columns."""
    qq = np.broadcast_to(q[:, None, :], (b, n * n, q.shape[1])).reshape(b * n * n, -1)
    return np.concatenate([h, qq], 1).astype(F32)


def log_softmax_np(z: np.ndarray) -> np.ndarray:
    m = z.max(1, keepdims=True)
    e = np.exp((z - m).astype(F32))
    return (z - m - np.log(e.sum(1, keepdims=True))).astype(F32)


def rl_forward_np(x, q, params, inject: int, dropout_mask=None, keep=True):
    """RelationalLayer.forward, model.py:104-162, in numpy fp32.

    params: dict with g_w (list of (out,in) arrays), g_b, f_w (3), f_b (3).
    dropout_mask: None (eval) or a (B, f_fc2) array already scaled by 1/(1-p)
    (model.py:158; nn.Dropout semantics).  Returns (log_probs, cache)."""
    b, n, k = x.shape
    L = len(params["g_w"])
    acts = []                                   # inputs of every g layer
    h = pair_matrix(x, q if inject == 0 else None)
    for l in range(L):
        if l == inject and l != 0:
            h = append_question(h, q, b, n)
        acts.append(h)
        h = np.maximum(h @ params["g_w"][l].T + params["g_b"][l], 0).astype(F32)   # :141-145
    hL = h
    x_g = hL.reshape(b, n * n, -1).sum(1, dtype=np.float64).astype(F32)                                # :151-152
    f1 = np.maximum(x_g @ params["f_w"][0].T + params["f_b"][0], 0).astype(F32)     # :155-156
    z2 = (f1 @ params["f_w"][1].T + params["f_b"][1]).astype(F32)                   # :157
    if dropout_mask is not None:
        z2 = (z2 * dropout_mask).astype(F32)                                        # :158
    f2 = np.maximum(z2, 0).astype(F32)                                              # :159
    z3 = (f2 @ params["f_w"][2].T + params["f_b"][2]).astype(F32)                   # :160
    out = log_softmax_np(z3)                                                        # :162
    cache = dict(acts=acts if keep else None, hL=hL, x_g=x_g, f1=f1, z2=z2, f2=f2, z3=z3,
                 out=out, dropout_mask=dropout_mask, inject=inject, shape=(b, n, k))
    return out, cache


def pair_features_np(x, q, params, inject: int, layer_idx: int):
    """extract.py:49-74 (hook on the INPUT of g_layers[layer_idx], extract.py:43,101): strip the question columns of an injection
    layer (:66-67), F.normalize(p=2, dim=2, eps=1e-12) per pair row (:68), max / mean over each question's n^2 pairs (:69-70).
    -> (maxf, avgf), each (B, F)."""
    b, n, k = x.shape
    _, cache = rl_forward_np(x, q, params, inject)
    z = cache["acts"][layer_idx]
    if layer_idx == inject:
        z = z[:, : z.shape[1] - q.shape[1]]
    z = z.reshape(b, n * n, -1)
    nrm = np.maximum(np.sqrt((z.astype(np.float64) ** 2).sum(2, keepdims=True)), 1e-12)
    u = (z / nrm).astype(F32)
    return u.max(1), u.mean(1, dtype=np.float64).astype(F32)


def rl_backward_np(x, q, params, cache, grad_out):
    """Gradients of RelationalLayer.forward w.r.t. x, q and every rl parameter,
    given d(loss)/d(log_probs) (B, A).  Mirrors what autograd derives for
    model.py:104-162."""
    b, n, k = cache["shape"]
    inject = cache["inject"]
    L = len(params["g_w"])
    Q = q.shape[1]
    out = cache["out"]
    g = {}
    # log_softmax backward: dz = g - softmax * sum(g)
    dz3 = (grad_out - np.exp(out) * grad_out.sum(1, keepdims=True)).astype(F32)
    g["f_w2"] = dz3.T @ cache["f2"]; g["f_b2"] = dz3.sum(0, dtype=np.float64).astype(F32)
    df2 = dz3 @ params["f_w"][2]
    dz2 = df2 * (cache["z2"] > 0)
    if cache["dropout_mask"] is not None:
        dz2 = dz2 * cache["dropout_mask"]
    dz2 = dz2.astype(F32)
    g["f_w1"] = dz2.T @ cache["f1"]; g["f_b1"] = dz2.sum(0, dtype=np.float64).astype(F32)
    df1 = (dz2 @ params["f_w"][1]) * (cache["f1"] > 0)
    df1 = df1.astype(F32)
    g["f_w0"] = df1.T @ cache["x_g"]; g["f_b0"] = df1.sum(0, dtype=np.float64).astype(F32)
    dxg = (df1 @ params["f_w"][0]).astype(F32)                    # (B, G)
    # sum backward = broadcast to all n*n pair rows
    dh = np.broadcast_to(dxg[:, None, :], (b, n * n, dxg.shape[1])).reshape(b * n * n, -1)
    dq = np.zeros_like(q, dtype=F32)
    h_out = cache["hL"]
    g_w, g_b = [None] * L, [None] * L
    for l in reversed(range(L)):
        a_in = cache["acts"][l]
        dz = (dh * (h_out > 0)).astype(F32)
        g_w[l] = (dz.T @ a_in).astype(F32)
        g_b[l] = dz.sum(0, dtype=np.float64).astype(F32)   # long reductions accumulate in fp64
        da = (dz @ params["g_w"][l]).astype(F32)
        if l == inject:
            dq += da[:, -Q:].reshape(b, n * n, Q).sum(1, dtype=np.float64).astype(F32)
            da = da[:, :-Q]
        dh = da
        h_out = a_in[:, : da.shape[1]] if l > 0 else None
    dp = dh.reshape(b, n, n, 2 * k)                               # [b,i,j,:]
    dx = dp[..., :k].sum(1, dtype=np.float64) + dp[..., k:].sum(2, dtype=np.float64)   # over i / over j
    g["g_w"], g["g_b"] = g_w, g_b
    return dx.astype(F32), dq.astype(F32), g


# --------------------------------------------------------------------------
# torch restatement (un-fused op sequence; autograd supplies the backward)
# --------------------------------------------------------------------------
if torch is not None:

    class ConvInputOracle(nn.Module):
        """model.py:9-36 -- 4 x [Conv2d(.,24,3,s=2,p=1) + BatchNorm2d + ReLU]."""

        def __init__(self):
            super().__init__()
            chans = [3, 24, 24, 24, 24]
            for i in range(4):
                setattr(self, "conv%d" % (i + 1), nn.Conv2d(chans[i], chans[i + 1], 3, stride=2, padding=1))
                setattr(self, "batchNorm%d" % (i + 1), nn.BatchNorm2d(24))

        def forward(self, img):
            x = img
            for i in range(1, 5):
                x = F.relu(getattr(self, "batchNorm%d" % i)(getattr(self, "conv%d" % i)(x)))
            return x

    class QuestionEmbedOracle(nn.Module):
        """model.py:39-58 -- Embedding(in+1, e) -> LSTM(batch_first) -> h_T."""

        def __init__(self, in_size, embed=32, hidden=128):
            super().__init__()
            self.wembedding = nn.Embedding(in_size + 1, embed)
            self.lstm = nn.LSTM(embed, hidden, batch_first=True)

        def forward(self, question):
            _, (h, _c) = self.lstm(self.wembedding(question))
            return h[0]

    class RelationalLayerOracle(nn.Module):
        """model.py:60-162 restated with the same un-fused op sequence
        (repeat / cat / Linear / relu / sum) so that it is also a fair CPU
        baseline for the reference's CPU path."""

        def __init__(self, in_size, out_size, qst_size, hyp):
            super().__init__()
            gl = hyp["g_layers"]
            self.f_fc1 = nn.Linear(gl[-1], hyp["f_fc1"])
            self.f_fc2 = nn.Linear(hyp["f_fc1"], hyp["f_fc2"])
            self.f_fc3 = nn.Linear(hyp["f_fc2"], out_size)
            self.dropout = nn.Dropout(p=hyp["dropout"])
            self.inject = hyp["question_injection_position"]
            self.in_size, self.qst_size, self.widths = in_size, qst_size, gl
            layers = []
            for idx, w in enumerate(gl):
                ins = in_size if idx == 0 else gl[idx - 1]
                layers.append(nn.Linear(ins + (qst_size if idx == self.inject else 0), w))
            self.g_layers = nn.ModuleList(layers)
            self.forced_dropout_mask = None       # tests: explicit (B,f_fc2) mask incl. 1/(1-p)

        def forward(self, x, qst):
            b, d, k = x.shape
            q4 = qst[:, None, None, :].expand(b, d, d, qst.shape[1])
            x_i = x[:, None, :, :].expand(b, d, d, k)
            x_j = x[:, :, None, :].expand(b, d, d, k)
            h = torch.cat([x_i, x_j], 3).reshape(b * d * d, 2 * k)
            for idx, layer in enumerate(self.g_layers):
                if idx == self.inject:
                    h = torch.cat([h.view(b, d, d, -1), q4], 3).reshape(b * d * d, -1)
                h = F.relu(layer(h))
            x_g = h.view(b, d * d, -1).sum(1)
            f = F.relu(self.f_fc1(x_g))
            f = self.f_fc2(f)
            if self.forced_dropout_mask is not None:
                f = f * self.forced_dropout_mask
            else:
                f = self.dropout(f)
            f = self.f_fc3(F.relu(f))
            return F.log_softmax(f, dim=1)

    class RNOracle(nn.Module):
        """model.py:164-223 restated (conv -> coord tag -> LSTM -> relational layer)."""

        def __init__(self, qdict_size, adict_size, hyp):
            super().__init__()
            self.state_desc = hyp["state_description"]
            self.conv = ConvInputOracle()
            self.text = QuestionEmbedOracle(qdict_size, embed=hyp["lstm_word_emb"], hidden=hyp["lstm_hidden"])
            self.rl = RelationalLayerOracle(hyp["rl_in_size"], adict_size, hyp["lstm_hidden"], hyp)

        def objects(self, img):
            if self.state_desc:
                return img
            x = self.conv(img)
            b, k, d, _ = x.shape
            lin = torch.linspace(-d / 2.0, d / 2.0, d, dtype=x.dtype, device=x.device)
            cx = lin[None, :].expand(d, d).reshape(1, 1, d * d).expand(b, 1, d * d)
            cy = lin[:, None].expand(d, d).reshape(1, 1, d * d).expand(b, 1, d * d)
            return torch.cat([x.view(b, k, d * d), cx, cy], 1).permute(0, 2, 1)

        def forward(self, img, qst_idxs):
            return self.rl(self.objects(img), self.text(qst_idxs))


def params_from_module(rl) -> dict:
    """Extract numpy params (nn.Linear (out,in) layout) from any module that
    exposes g_layers / f_fc1..3 like model.py:64-66,:89-101."""
    t = lambda p: p.detach().cpu().numpy().astype(F32)
    return dict(
        g_w=[t(l.weight) for l in rl.g_layers], g_b=[t(l.bias) for l in rl.g_layers],
        f_w=[t(rl.f_fc1.weight), t(rl.f_fc2.weight), t(rl.f_fc3.weight)],
        f_b=[t(rl.f_fc1.bias), t(rl.f_fc2.bias), t(rl.f_fc3.bias)],
    )
