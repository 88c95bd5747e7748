"""Shared helpers: load golden fixtures and rebuild their closed-form inputs."""
import json
import os

import numpy as np

from oracle import formula

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(tag):
    z = np.load(os.path.join(GOLD, tag + ".npz"), allow_pickle=False)
    d = {k: z[k] for k in z.files}
    d["meta"] = json.loads(str(d["meta"]))
    return d


def rl_case(meta):
    """(hyp, state(np), x, q, labels) for a record_rl fixture (see make_golden.py)."""
    hyp = formula.HYP[meta["cfg"]]
    b, n, seed = meta["b"], meta["n"], meta["seed"]
    k, Q = hyp["rl_in_size"] // 2, hyp["lstm_hidden"]
    sd = formula.formula_rl_state(hyp, seed)
    x = formula.formula_objects(b, n, k, seed + 1, from_pixels=not hyp["state_description"])
    q = formula.hash_uniform((b, Q), seed + 2, -1.0, 1.0)
    lab = formula.hash_ints((b,), seed + 3, 0, formula.ADICT)
    return hyp, sd, x, q, lab


def nll_grad(log_probs, labels):
    """d mean-NLL / d log_probs  (train.py:41)."""
    g = np.zeros_like(log_probs)
    g[np.arange(len(labels)), labels] = -1.0 / len(labels)
    return g


def rel_err(a, b):
    """max-norm relative error, the metric the 1e-3 parity bar is written in."""
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def check_grads(gold, named_grads, tol, report=None):
    """Compare {name: array} against a fixture's grad/, gradsample/, gradnorm/ entries."""
    worst = 0.0
    for key in gold:
        if key.startswith("grad/"):
            name = key[5:]
            e = rel_err(named_grads[name], gold[key])
        elif key.startswith("gradsample/"):
            name = key[len("gradsample/"):]
            got = np.asarray(named_grads[name]).reshape(-1)[gold["gradsample_idx/" + name]]
            scale = float(gold["gradnorm/" + name]) / np.sqrt(np.asarray(named_grads[name]).size)
            e = float(np.abs(got.astype(np.float64) - gold[key]).max() / max(np.abs(gold[key]).max(), scale, 1e-30))
            en = abs(float(np.linalg.norm(np.asarray(named_grads[name], dtype=np.float64))) - float(gold["gradnorm/" + name])) \
                / max(float(gold["gradnorm/" + name]), 1e-30)
            e = max(e, en)
        else:
            continue
        if report is not None:
            report[name] = e
        worst = max(worst, e)
        assert e <= tol, "%s: rel err %.3e > %.1e" % (name, e, tol)
    return worst
