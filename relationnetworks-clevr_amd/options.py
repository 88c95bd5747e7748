"""Execution options of the MI355X path -- ONE explicit object, read from the process environment ONCE, at import.

Nothing on the hot path looks at os.environ: dispatch depends on `OPT` only, so a call behaves the same whatever the
environment does afterwards.  Fifteen entries are left (round 3 had 31: the ones that only kept an older implementation alive for
A/B history went with those implementations): the arithmetic mode, the two numerically visible choices of the chain path (e4m3
copies, on-chip pair reductions) that tests pin against their alternatives, and the trainer's launch structure.  Every switch
defaults to the fast path (tests use `override(...)`, not the environment).  The C library has no
options of its own: librn_hip.so never reads the environment (kernel-variant knobs exist in RN_DIAG builds only)."""
from __future__ import annotations

import contextlib
import os

# attribute -> (environment variable, default, meaning)
_SPEC = {
    "precision":         ("RN_PRECISION", "auto", 'arithmetic mode of modules whose hyp has no "precision": auto | f16s | bf16 | fp32 | bf16x3'),
    "eval_two_pass":     ("RN_NO_EVAL_TWO_PASS", True, "eval() without gradients on the chain path: hi + lo split weights on every g layer instead of the tile-dithered single pass (log-probs independent of batch position / object order; ~1.6x the forward chain's time)"),
    "h8":                ("RN_H8", True, "e4m3 copies of H_0..2 for the weight gradients (False: bf16 copies, the last layer's dZ stored; what the trainer's copy guard falls back to)"),
    "chain_reduce":      ("RN_NO_CHAIN_REDUCE", True, "pair-axis reductions of layer 0's gradient inside the backward chain (dZ_0 never stored; any chain shape, the padded j axis included)"),
    "wgrad_overlap":     ("RN_NO_WGRAD_OVERLAP", True, "weight gradients on side streams (bench.py switches it off to time every kernel alone)"),
    "overlap_streams":   ("RN_OVERLAP_STREAMS", True, "question encoder beside the conv stack (second stream)"),
    "fphi_split":        ("RN_NO_FPHI_SPLIT", True, "f_phi as the feature-split fp32 MFMA chain in one launch (rn_f_phi_split: B <= 64, 256-wide layers); False: the row-split FMA kernels"),
    "fphi_fused_bwd":    ("RN_NO_FPHI_FUSED_BWD", True, "trainer: f_phi's backward dz chain in the forward launch (the loss gradient is the cached 1)"),
    "fused_loss":        ("RN_NO_FUSED_LOSS", True, "loss inside the f_phi launch (trainer)"),
    "grads_in_bucket":   ("RN_NO_GRADS_IN_BUCKET", True, "backward kernels write parameter gradients straight into a registered FlatGradBucket (trainer)"),
    "fused_adam":        ("RN_NO_FUSED_ADAM", True, "fused clip + Adam on the flat bucket (trainer)"),
    "graph_allreduce":   ("RN_NO_GRAPH_ALLREDUCE", True, "N > 1 over RCCL: the gradient all-reduce inside the captured step too (else eager, with the optimiser behind it)"),
    "native_dropout":    ("RN_NO_NATIVE_DROPOUT", True, "f_phi dropout mask from the library's counter-based generator (device-side draw counter: a replayed step graph needs no generator fills in front of it); False: torch's F.dropout"),
    "graph_adam":        ("RN_NO_GRAPH_ADAM", True, "clip + Adam (and, N > 1, the all-reduce) inside the captured step"),
    "dp_timeout":        ("RN_DP_TIMEOUT", 300, "N > 1: seconds a trainer waits on other ranks / on the first in-graph exchange before the process gives up (exit 124); 0 = for ever"),
}


def _read(env, default, environ):
    raw = environ.get(env)
    if raw is None or raw == "":
        return default
    if isinstance(default, bool):
        if env.startswith("RN_NO_"):
            return raw != "1"                      # RN_NO_X=1 switches X off
        return raw != "0"                          # RN_X=0 switches X off
    if isinstance(default, int):
        return int(raw)
    return raw


class Options:
    __slots__ = tuple(_SPEC)

    def __init__(self, environ=None):
        environ = os.environ if environ is None else environ
        for name, (env, default, _doc) in _SPEC.items():
            setattr(self, name, _read(env, default, environ))

    def as_dict(self):
        return {k: getattr(self, k) for k in _SPEC}

    def non_default(self):
        return {k: getattr(self, k) for k, (_e, d, _doc) in _SPEC.items() if getattr(self, k) != d}


OPT = Options()


@contextlib.contextmanager
def override(**kw):
    """Temporarily change options (tests, A/B tools): `with override(h8=False): ...`"""
    old = {k: getattr(OPT, k) for k in kw}
    try:
        for k, v in kw.items():
            if k not in _SPEC:
                raise AttributeError("unknown option %r" % k)
            setattr(OPT, k, v)
        yield OPT
    finally:
        for k, v in old.items():
            setattr(OPT, k, v)
