"""Execution options of the MI355X path -- ONE explicit object, read from the process environment ONCE, at import.

Nothing on the hot path looks at os.environ: dispatch depends on `OPT` only, so a call behaves the same whatever the
environment does afterwards.  Every switch defaults to the fast path; the RN_NO_* variables exist for A/B measurements and
for the tests that pin one path against another (tests use `override(...)`, not the environment).  The C library has no
options of its own: librn_hip.so never reads the environment (kernel-variant knobs exist in RN_DIAG builds only)."""
from __future__ import annotations

import contextlib
import os

# attribute -> (environment variable, default, meaning)
_SPEC = {
    "precision":         ("RN_PRECISION", "auto", 'arithmetic mode of modules whose hyp has no "precision": auto | f16s | bf16 | fp32'),
    "h8":                ("RN_H8", True, "e4m3 copies of H_0..2 for the weight gradients (False: 16-bit copies, the last layer's dZ stored)"),
    "rr_chain":          ("RN_NO_RR_CHAIN", True, "register-resident g_theta chains (rn_chain_rr.hip)"),
    "rr_masks":          ("RN_NO_RR_MASKS", True, "... with ReLU lane masks instead of a stored last activation"),
    "inj_chain":         ("RN_NO_INJ_CHAIN", True, "... with the question injected at layer 2 as a bias row (ir-*)"),
    "fused_chain":       ("RN_NO_FUSED_CHAIN", True, "LDS-resident fused chain (rn_chain.hip) where the register-resident one does not apply"),
    "fused_bwd":         ("RN_NO_FUSED_BWD", True, "fused backward chain"),
    "algebraic_fwd0":    ("RN_NO_ALGEBRAIC_FWD0", True, "first layer factored through the pair structure (tables instead of the pair matrix)"),
    "algebraic_wgrad0":  ("RN_NO_ALGEBRAIC_WGRAD0", True, "layer-0 weight gradient from the pair reductions"),
    "gated_wgrad":       ("RN_NO_GATED_WGRAD", True, "last layer's gradient never stored (gate job of rn_g_wgrad_blocked)"),
    "chain_reduce":      ("RN_NO_CHAIN_REDUCE", True, "pair-axis reductions of layer 0's gradient inside the backward chain (dZ_0 never stored; n % 32 == 0)"),
    "gate_fwd":          ("RN_NO_GATE_FWD", True, "... and its gate image written by the f16s forward chain (else by rn_relu_gate_image in the backward pass)"),
    "rq_from_wgrad":     ("RN_NO_RQ_FROM_WGRAD", True, "injected layer's per-question sums from the weight-gradient kernel's db partials"),
    "fused_pair_tail":   ("RN_NO_FUSED_PAIR_TAIL", True, "dx and dq in one launch (rn_pair_dx_dq), straight into the conv grid's layout"),
    "fused_pair_sum":    ("RN_NO_FUSED_PAIR_SUM", True, "pair-sum partials added inside the f_phi launch"),
    "grid_fast":         ("RN_NO_GRID_FAST", True, "kernels take the conv grid + coordinate table (no concatenated object tensor)"),
    "pack_ahead":        ("RN_NO_PACK_AHEAD", True, "weight images packed on the question encoder's side stream"),
    "wgrad_overlap":     ("RN_NO_WGRAD_OVERLAP", True, "weight gradients on a side stream"),
    "wgrad0_stream":     ("RN_WGRAD0_STREAM", 2, "layer-0 weight gradient (from the pair reductions): 0 on the weight-gradient stream, 1 on the main stream behind dx / dq, 2 on a stream of its own (measured, one box: 85.6 / 87.0 / 87.9 k q/s)"),
    "wgrad_late":        ("RN_WGRAD_LATE", 0, "1 / 2: start the weight-gradient stream after the pair reduction / after dx, dq (measured slower)"),
    "direct_conv":       ("RN_NO_DIRECT_CONV", True, "own 3x3 / stride-2 convolution kernels (else MIOpen)"),
    "direct_conv_wgrad": ("RN_NO_DIRECT_CONV_WGRAD", True, "... and their weight gradient"),
    "lstm_tail_fused": ("RN_NO_FUSED_LSTM_TAIL", True, "question encoder backward: embedding gradient first, bias gradients in its launch"),
    "batch_copy_fused": ("RN_NO_BATCH_COPY_FUSED", True, "captured step: the batch's three hand-off copies as one launch"),
    "bn_wgrad_fused": ("RN_NO_BN_WGRAD_FUSED", True, "first conv block: batch-norm backward pass 2 inside the weight-gradient kernel"),
    "fused_bn":          ("RN_NO_FUSED_BN", True, "fused conv-bias + BatchNorm + ReLU kernels"),
    "fused_lstm":        ("RN_NO_FUSED_LSTM", True, "one-launch embedding + LSTM"),
    "overlap_streams":   ("RN_OVERLAP_STREAMS", True, "question encoder beside the conv stack (second stream)"),
    "fused_nll":         ("RN_NO_FUSED_NLL", True, "own NLL kernels"),
    "fphi_fused_bwd":    ("RN_NO_FPHI_FUSED_BWD", True, "trainer: f_phi's backward dz chain in the forward launch (the loss gradient is the cached 1)"),
    "fused_loss":        ("RN_NO_FUSED_LOSS", True, "loss inside the f_phi launch (trainer)"),
    "grads_in_bucket":   ("RN_NO_GRADS_IN_BUCKET", True, "backward kernels write parameter gradients straight into a registered FlatGradBucket (trainer)"),
    "fused_adam":        ("RN_NO_FUSED_ADAM", True, "fused clip + Adam on the flat bucket (trainer)"),
    "graph_allreduce":   ("RN_NO_GRAPH_ALLREDUCE", True, "N > 1 over RCCL: the gradient all-reduce inside the captured step too (else eager, with the optimiser behind it)"),
    "graph_adam":        ("RN_NO_GRAPH_ADAM", True, "... inside the captured step on one GPU"),
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
