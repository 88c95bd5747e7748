"""Closed-form, torch-independent test data  --  TEST INFRASTRUCTURE ONLY.

Golden fixtures store only *outputs*; inputs and weights are regenerated from
an integer hash (splitmix64) so that they are bit-identical on every machine
and independent of torch's RNG stream."""
from __future__ import annotations

import json
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(z: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = (z + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def hash_uniform(shape, seed: int, lo: float = -1.0, hi: float = 1.0) -> np.ndarray:
    """fp32 array, U[lo,hi), element i = splitmix64(seed*2^32 + i)."""
    n = int(np.prod(shape))
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64) + (np.uint64(seed) << np.uint64(32))
    u = (_splitmix64(idx) >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


def hash_ints(shape, seed: int, lo: int, hi: int) -> np.ndarray:
    """int64 array, uniform in [lo, hi)."""
    n = int(np.prod(shape))
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64) + (np.uint64(seed) << np.uint64(32))
    r = _splitmix64(idx) % np.uint64(hi - lo)
    return (r.astype(np.int64) + lo).reshape(shape)


# Hyper-parameter sets of the reference (config.json:3-58), restated as data so
# the GPU box (which has no /root/reference) can rebuild the same models.
HYP = {
    "original-fp": dict(state_description=False, g_layers=[256, 256, 256, 256], question_injection_position=0,
                        f_fc1=256, f_fc2=256, dropout=0.5, lstm_hidden=128, lstm_word_emb=32, rl_in_size=52),
    "original-sd": dict(state_description=True, g_layers=[512, 512, 512, 512], question_injection_position=0,
                        f_fc1=512, f_fc2=1024, dropout=0.05, lstm_hidden=256, lstm_word_emb=32, rl_in_size=14),
    "ir-fp": dict(state_description=False, g_layers=[256, 256, 256, 256], question_injection_position=2,
                  f_fc1=256, f_fc2=256, dropout=0.5, lstm_hidden=128, lstm_word_emb=32, rl_in_size=52),
    "ir-sd": dict(state_description=True, g_layers=[512, 512, 512, 512], question_injection_position=2,
                  f_fc1=512, f_fc2=1024, dropout=0.05, lstm_hidden=256, lstm_word_emb=32, rl_in_size=14),
}
ADICT, QDICT = 28, 82          # vocabulary sizes implied by the released checkpoints


def rl_layer_shapes(hyp, adict=ADICT):
    """[(name, (out,in))] in the reference's parameter creation order for `rl`
    (model.py:64-66 then :91-100)."""
    gl, inj, Q = hyp["g_layers"], hyp["question_injection_position"], hyp["lstm_hidden"]
    shapes = [("f_fc1", (hyp["f_fc1"], gl[-1])), ("f_fc2", (hyp["f_fc2"], hyp["f_fc1"])), ("f_fc3", (adict, hyp["f_fc2"]))]
    for i, w in enumerate(gl):
        ins = (hyp["rl_in_size"] if i == 0 else gl[i - 1]) + (Q if i == inj else 0)
        shapes.append(("g_layers.%d" % i, (w, ins)))
    return shapes


def formula_rl_state(hyp, seed: int, adict=ADICT, gain: float = 1.0):
    """state_dict-style {name: fp32 array} for the relational layer with
    U(+-gain/sqrt(fan_in)) weights and biases (nn.Linear's default range)."""
    sd = {}
    for li, (name, (o, i)) in enumerate(rl_layer_shapes(hyp, adict)):
        bound = gain / np.sqrt(i)
        sd[name + ".weight"] = hash_uniform((o, i), seed * 100 + 2 * li, -bound, bound)
        sd[name + ".bias"] = hash_uniform((o,), seed * 100 + 2 * li + 1, -bound, bound)
    return sd


def params_from_state(sd, n_g: int) -> dict:
    """numpy param dict in the layout rn_oracle.rl_forward_np expects."""
    return dict(
        g_w=[sd["g_layers.%d.weight" % i] for i in range(n_g)], g_b=[sd["g_layers.%d.bias" % i] for i in range(n_g)],
        f_w=[sd["f_fc%d.weight" % i] for i in (1, 2, 3)], f_b=[sd["f_fc%d.bias" % i] for i in (1, 2, 3)],
    )


def formula_objects(b, n, k, seed, from_pixels=True, d=None):
    """Objects x (B,n,k).  from_pixels: channels 0..k-3 are post-ReLU conv
    features (>=0, ~half zeros like a ReLU output), the last two are the
    coordinate tags of model.py:208-218.  Otherwise: state-description style
    rows with trailing zero-padded objects (utils.py:101-107)."""
    from .rn_oracle import coord_table
    if from_pixels:
        d = d or int(round(np.sqrt(n)))
        assert d * d == n
        feat = np.maximum(hash_uniform((b, n, k - 2), seed, -1.0, 1.5), 0)
        lin = coord_table(d)
        p = np.arange(n)
        cx, cy = lin[p % d], lin[p // d]
        x = np.concatenate([feat, np.broadcast_to(cx[None, :, None], (b, n, 1)),
                            np.broadcast_to(cy[None, :, None], (b, n, 1))], 2)
        return np.ascontiguousarray(x, dtype=np.float32)
    x = hash_uniform((b, n, k), seed, -3.0, 3.0)
    real = [3, 6, 10, 12]
    for bi in range(b):
        x[bi, real[bi % 4] if n == 12 else max(1, n - bi % 3):, :] = 0.0
    return x


def load_config_json():
    """The package's own config.json (same schema as the reference's)."""
    p = os.path.join(_HERE, "..", "relationnetworks-clevr_amd", "config.json")
    with open(p) as f:
        return json.load(f)["hyperparams"]


def formula_fill_state(shapes: dict, seed: int) -> dict:
    """Closed-form values for a *full* RN state_dict (conv + BN + embedding +
    LSTM + rl), keyed by the reference's parameter names.
    `shapes` maps name -> shape in state_dict order."""
    out = {}
    for idx, (name, shape) in enumerate(shapes.items()):
        shape = tuple(int(s) for s in shape)
        s = seed * 1000 + idx
        if name.endswith("num_batches_tracked"):
            out[name] = np.zeros(shape, dtype=np.int64)
        elif name.endswith("running_var"):
            out[name] = hash_uniform(shape, s, 0.5, 1.5)
        elif name.endswith("running_mean"):
            out[name] = hash_uniform(shape, s, -0.2, 0.2)
        elif "batchNorm" in name and name.endswith("weight"):
            out[name] = hash_uniform(shape, s, 0.5, 1.5)
        elif "wembedding" in name:
            out[name] = hash_uniform(shape, s, -1.0, 1.0)
        else:
            fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else int(shape[0])
            bound = 1.0 / np.sqrt(max(fan_in, 1))
            if "lstm" in name:
                bound = 1.0 / np.sqrt(shape[-1] if "hh" in name or len(shape) == 1 else shape[-1])
            out[name] = hash_uniform(shape, s, -bound, bound)
    return out
