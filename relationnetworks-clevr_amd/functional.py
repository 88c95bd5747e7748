"""Relational-layer hot path (model.py:104-162 of the reference) as ONE
torch.autograd.Function over the C-ABI HIP kernels.

forward : K1 pair build -> g_theta GEMM chain (bias+ReLU fused) -> pair sum -> f_phi -> log_softmax
backward: log_softmax/f_phi grads -> pair-sum broadcast + ReLU gate -> per layer {wgrad, dgrad+gate}
          -> pair-axis reductions (algebraic dP) -> dx, dq

PyTorch only allocates buffers and supplies the stream; all arithmetic runs in
librn_hip.so.  There is no CPU / eager fallback."""
from __future__ import annotations

import os
import torch

from . import rn_hip as H
from .options import OPT

PRECISIONS = ("bf16", "f16s", "fp32")


def _ru(v, m):
    return (v + m - 1) // m * m


class LayerPlan:
    """Static shape bookkeeping of the g_theta chain for one (n, k, Q, widths, inject)."""

    def __init__(self, k, Q, widths, inject):
        self.k, self.Q, self.widths, self.inject = k, Q, list(widths), inject
        self.L = len(widths)
        self.ktrue, self.kpad, self.qcol = [], [], []
        for l, w in enumerate(widths):
            base = 2 * k if l == 0 else widths[l - 1]
            kt = base + (Q if l == inject else 0)
            self.ktrue.append(kt)
            self.kpad.append(_ru(kt, 64))
            self.qcol.append(base if l == inject else -1)
        for w in widths:
            if w % 256:
                raise RuntimeError("g_layers widths must be multiples of 256 for the MI355X kernels (got %r)" % (widths,))
        if Q % 8:
            raise RuntimeError("lstm_hidden must be a multiple of 8 (got %d)" % Q)

    def in_ld(self, l):
        """leading dimension of the input buffer of layer l (== kpad[l])."""
        return self.kpad[l]


class PackedWeights:
    """bf16/fp32 MFMA-operand copies of the g weights: forward (N, Kpad) and, for l >= 1,
    transposed (G_{l-1}, N) for dgrad.  Re-packed only when a weight's version counter moves."""

    def __init__(self):
        self.key = None
        self.fwd, self.bwd = [], []
        self.hi, self.lo = [], []                  # fp16 split copies for the f16s forward
        self.fT = None                             # transposed fp32 copies of the f_phi weights
        self.w0T = None                            # W_0^T in fp32: the table kernel of the factored first layer
        self._last = None                          # arguments of the previous get(): what repack_ahead() repeats
        self._ahead = None                         # key of an ahead-of-time pack not consumed yet
        self.frag_hi, self.frag_lo = [], []        # fragment-major fp16 hi / lo images (f16s on the register-resident chain)
        self.frag = []                             # fragment-major copies for the register-resident chains:
        self.fragT = []                            #   forward W_l, backward step s -> W_{L-1-s}^T

    def get(self, plan: LayerPlan, g_w, code, split=False, bwd_images=True, rr_only=False, f_w=None, alg0_k=0, inj=0):
        """rr_only: the call is known to run the register-resident chains in both directions -- only their
        fragment-major images are packed (the row-major copies feed the other kernels).  alg0_k > 0: the forward
        chain runs with the factored first layer (rn_g_chain_fwd_rr_alg0): the layer-0 image holds W0[:, 0:k] only and
        W0^T is kept in fp32 for the table kernel (self.w0T).  inj > 0: the chain runs with the question injected at layer
        `inj` as a bias row (inj_chain_ok): that layer's image holds W[:, 0:256] (the H columns) only."""
        key = (code, split, bwd_images, rr_only, alg0_k, inj, tuple((w.data_ptr(), w._version) for w in list(g_w) + list(f_w or ())))
        self._last = (plan, tuple(g_w), code, split, bwd_images, rr_only, tuple(f_w) if f_w is not None else None, alg0_k, inj)
        if self._ahead == key:                              # packed by repack_ahead() earlier in this forward pass
            self._ahead = None
            return self.fwd, self.bwd
        self._ahead = None
        # while a hipGraph is being captured the pack kernels must be part of it (a replay sees
        # new weights every step), so the cache is bypassed
        if key == self.key and not torch.cuda.is_current_stream_capturing():
            return self.fwd, self.bwd
        return self._pack(key, plan, g_w, code, split, bwd_images, rr_only, f_w, alg0_k, inj)

    def repack_ahead(self):
        """Repeat the previous get()'s pack NOW, on the caller's current stream -- RN.forward calls this on the question
        encoder's side stream, whose fork and join around the conv stack exist anyway (a fork / join of its own costs
        more than the 11 us it hides: measured).  The next get() with the same arguments returns the images without
        launching anything; the caller's join orders it after this stream."""
        if self._last is None or not OPT.pack_ahead:
            return
        plan, g_w, code, split, bwd_images, rr_only, f_w, alg0_k, inj = self._last
        if not torch.is_grad_enabled():
            bwd_images = False
        key = (code, split, bwd_images, rr_only, alg0_k, inj, tuple((w.data_ptr(), w._version) for w in list(g_w) + list(f_w or ())))
        if key != self.key or torch.cuda.is_current_stream_capturing():
            self._pack(key, plan, g_w, code, split, bwd_images, rr_only, f_w, alg0_k, inj)
        self._ahead = key

    def _pack(self, key, plan, g_w, code, split, bwd_images, rr_only, f_w, alg0_k, inj=0):
        dt = H.torch_dtype(code)
        dev = g_w[0].device
        self.fwd, self.bwd, self.hi, self.lo, self.frag, self.fragT = [], [], [], [], [], []
        self.frag_hi, self.frag_lo = [], []
        rr = rr_chain_ok(plan, code) or inj > 0
        frag_jobs = []
        for l, w in enumerate(g_w):
            N, kt = w.shape
            assert kt == plan.ktrue[l] and N == plan.widths[l], (w.shape, plan.ktrue[l], plan.widths[l])
            wc = w.detach()
            if not wc.is_contiguous():
                wc = wc.contiguous()
            # columns that enter the MFMA image: the factored first layer keeps W0[:, 0:k], an injected layer W[:, 0:G_prev]
            kimg = alg0_k if (l == 0 and alg0_k) else (plan.widths[l - 1] if (inj and l == inj) else kt)
            if rr and rr_only:
                self.fwd.append(None)
            else:
                wp = torch.empty(N, plan.kpad[l], dtype=dt, device=dev)
                H.pack_matrix(wc, kt, 1, N, kt, wp, code, plan.kpad[l], N)
                self.fwd.append(wp)
            if rr and split:
                # f16s on the register-resident chains: layer 0 = hi + lo images (two passes), layers >= 1 = F16S_DITHER
                # tile-dithered hi images (one pass; include/rn_hip.h, rn_g_chain_fwd_rr_f16s)
                V = 1 if l == 0 else H.F16S_DITHER
                wh = torch.empty(V, 256 * 256, dtype=torch.float16, device=dev)
                frag_jobs.append((wc, kt, 1, N, kimg, wh, 4 | int(l == 0) | ((V << 8) if V > 1 else 0)))
                self.frag_hi.append(wh)
                if l == 0:
                    wl = torch.empty(256 * 256, dtype=torch.float16, device=dev)
                    frag_jobs.append((wc, kt, 1, N, kimg, wl, 8 | 1))
                    self.frag_lo.append(wl)
            elif rr:
                wf = torch.empty(256 * 256, dtype=dt, device=dev)
                frag_jobs.append((wc, kt, 1, N, kimg, wf, l == 0))
                self.frag.append(wf)
            if split and not (rr and rr_only):
                hi = torch.empty(N, plan.kpad[l], dtype=torch.float16, device=dev)
                lo = torch.empty(N, plan.kpad[l], dtype=torch.float16, device=dev)
                H.pack_matrix_split(wc, kt, 1, N, kt, hi, lo, plan.kpad[l], N)
                self.hi.append(hi)
                self.lo.append(lo)
            if l >= 1 and not (rr and rr_only):
                gp = plan.widths[l - 1]           # only the H_{l-1} columns take part in dgrad
                wt = torch.empty(gp, N, dtype=dt, device=dev)
                H.pack_matrix(wc, 1, kt, gp, N, wt, code, N, gp)      # wt[k][n] = w[n][k]
                self.bwd.append(wt)
            else:
                self.bwd.append(None)
        if rr and bwd_images:
            self.fragT = list(torch.empty(plan.L - 1, 256 * 256, dtype=dt, device=dev))     # equally spaced (rn_g_chain_bwd_rr)
            for st, wf in enumerate(self.fragT):
                wc = g_w[plan.L - 1 - st].detach().contiguous()
                frag_jobs.append((wc, 1, wc.shape[1], 256, 256, wf, st == 0))      # element (in, out) = W[out][in]
        self.w0T = None
        if rr and alg0_k:
            w0 = g_w[0].detach().contiguous()
            self.w0T = torch.empty(w0.shape[1], w0.shape[0], dtype=torch.float32, device=dev)
            frag_jobs.append((w0, w0.shape[1], 1, w0.shape[0], w0.shape[1], self.w0T, 2))
        # f_phi weights as (in, out) fp32 copies: the forward kernel's thread-per-output-feature walk is then coalesced
        self.fT = None
        if f_w is not None and all(max(w.shape) <= 256 for w in f_w):
            self.fT = [torch.empty(w.shape[1], w.shape[0], dtype=torch.float32, device=dev) for w in f_w]
            for w, wt in zip(f_w, self.fT):
                wc = w.detach().contiguous()
                frag_jobs.append((wc, wc.shape[1], 1, wc.shape[0], wc.shape[1], wt, 2))
        if frag_jobs:
            H.pack_matrix_frag_many(frag_jobs)                                     # one launch for all images
        self.key = key
        return self.fwd, self.bwd


_SIDE_STREAMS = {}


def _side_stream(dev, which=0):
    s = _SIDE_STREAMS.get((dev, which))
    if s is None:
        s = _SIDE_STREAMS[(dev, which)] = torch.cuda.Stream(device=dev)
    return s


# ---- gradients straight into a FlatGradBucket (dp.py) ---------------------------------------------------------------
# A bucket registers (parameter -> slot of its flat buffer).  A backward function that is about to allocate the gradient of a
# registered parameter takes a FRESH VIEW of the slot instead and lets its kernels write there: autograd then assigns that
# view as .grad (no copy: a fresh view is uniquely owned), and the bucket's gather step finds the gradient already in place --
# no concatenation kernel between the end of the backward pass and the optimiser.  Only when autograd will ASSIGN
# (_assign_only: .grad is None, no hooks): accumulating `grad += view-of-the-same-memory` would double the new gradient.
_GRAD_SLOTS = {}


def register_grad_slots(params, flat, offsets):
    import weakref
    for p_, o in zip(params, offsets):
        _GRAD_SLOTS[id(p_)] = (weakref.ref(p_), weakref.ref(flat), o)


def grad_out(param, shape=None):
    """The tensor a backward function writes `param`'s gradient into (see above); shape: the gradient's shape when the caller
    needs another one than the parameter's (then no slot is used unless the element count matches a contiguous view)."""
    shape = tuple(param.shape) if shape is None else tuple(shape)
    slot = _GRAD_SLOTS.get(id(param)) if OPT.grads_in_bucket else None
    if slot is not None:
        pref, fref, off = slot
        flat = fref()
        if pref() is param and flat is not None and flat.device == param.device and shape == tuple(param.shape) and _assign_only(param):
            return flat[off:off + param.numel()].view(shape)
    return torch.empty(shape, dtype=torch.float32, device=param.device)


# ---- the trainer's loss gradient ------------------------------------------------------------------------------------
# loss.backward() hands the first backward function a gradient of 1.  The trainer passes THIS cached tensor (no fill launch),
# and a function that finds it knows the value without reading device memory: the f_phi forward launch may then have run the
# backward dz chain already (rn_f_phi_fwd_bwd_from_partials).  Any other tensor takes the general path.
_UNIT_LOSS_GRAD = {}


def unit_loss_grad(device, dtype=torch.float32):
    key = (torch.device(device), dtype)
    t = _UNIT_LOSS_GRAD.get(key)
    if t is None:
        t = _UNIT_LOSS_GRAD[key] = torch.ones((), dtype=dtype, device=device)
    return t


def is_unit_loss_grad(g):
    t = _UNIT_LOSS_GRAD.get((g.device, g.dtype)) if g is not None else None
    return t is not None and g.data_ptr() == t.data_ptr() and g.numel() == 1


# Set to a list by dp.DataParallelTrainer.check_activation_copies(): the next training forward appends (layer, counters) of every
# e4m3 activation copy it writes (rn_fp8_copy_health).  None: no probe, nothing is launched.
COPY_HEALTH_PROBE = None


class RRMasks:
    """ReLU lane masks of the register-resident forward chain: what the backward pass keeps INSTEAD of the last
    activation (rn_g_chain_fwd_rr / rn_g_chain_bwd_rr)."""

    def __init__(self, masks, gate=None):
        self.masks = masks
        self.gate = gate          # the last layer's gate as an e4m3 {0, 1} row-blocked image, when the forward chain wrote it


def rr_chain_ok(plan: LayerPlan, code):
    """The register-resident forward chain (rn_chain_rr.hip): bf16, exactly four 256-wide g layers, question
    injected at layer 0 with a padded layer-0 reduction length of 192 or 256 (the headline shape family)."""
    if not OPT.rr_chain:
        return False
    return (code == H.RN_BF16 and plan.L == 4 and all(w == 256 for w in plan.widths) and plan.kpad[0] in (192, 256)
            and all(kp == 256 for kp in plan.kpad[1:]))


def fused_chain_ok(plan: LayerPlan, code, B, n):
    """The fused LDS-resident chain (rn_chain.hip) covers the headline shape family: bf16 storage,
    all g widths 256, question injected at layer 0 (so every later layer has K == 256), and whole
    128-row tiles per question.  Everything else runs the per-layer kernels."""
    if not OPT.fused_chain:
        return False
    return (code == H.RN_BF16 and all(w == 256 for w in plan.widths) and plan.kpad[0] <= 256
            and all(kp == 256 for kp in plan.kpad[1:]) and (B * n * n) % H.g_chain_tile() == 0 and plan.L <= 8)


def inj_chain_ok(plan: LayerPlan, code, n, k, M):
    """The register-resident chains with the question injected at layer 2 (the reference's "IR" variants, config.json
    ir-fp): the factored first layer without a question term + the per-question bias row W_2[:, 256:] q[b] + b_2 at layer 2
    (rn_g_chain_fwd_rr*_alg0 with inject_layer = 2).  Four 256-wide layers, whole waves per (question, i), whole 256-row
    tiles per question."""
    if not OPT.rr_chain or not OPT.rr_masks or not OPT.inj_chain:
        return False
    return (code == H.RN_BF16 and plan.L == 4 and all(w == 256 for w in plan.widths) and plan.inject == 2 and k <= 32
            and plan.ktrue[2] == 256 + plan.Q and n % 32 == 0 and (n * n) % H.g_chain_rr_tile() == 0 and M % H.g_chain_rr_tile() == 0)


def padded_j(n):
    """Pair rows per (question, i) group on the factored-first-layer paths: n itself when a wave's 32 rows fit (n % 32 == 0), else
    the next multiple of 32 -- the f16s chain then runs on a PADDED j axis (include/rn_hip.h, rn_g_chain_fwd_rr_f16s_alg0; the
    14 x 14 grid: 196 -> 224) -- or None when that kernel does not cover n either (n % 4 != 0)."""
    if n % 32 == 0:
        return n
    return _ru(n, 32) if n % 4 == 0 else None


def f16s_ok(plan: LayerPlan, B, n):
    """Shapes the "f16s" arithmetic (fp16 activations x split / tile-dithered fp16 weights) has a kernel for."""
    return (fused_chain_ok(plan, H.RN_BF16, B, n) or inj_chain_ok(plan, H.RN_BF16, n, plan.k, B * n * n)
            or alg0_forward_ok(plan, H.RN_BF16, n, plan.k, B * n * n, f16s=True))


def _h_copy_dtype(plan, dt, M):
    """Storage type of the H_0..2 copies the register-resident chains keep for the weight gradient (row-blocked images,
    rn_g_wgrad_blocked is their only reader): OCP e4m3 bytes -- half the bytes written by the forward chain and read back --
    or the chain's 16-bit type with RN_H8=0 (A/B measurements, error comparisons)."""
    if OPT.h8 and dt == torch.bfloat16 and all(w == 256 for w in plan.widths):
        return torch.float8_e4m3fn
    return dt


def alg0_wgrad_ok(plan, k):
    """Layer-0 weight gradient from the pair reductions (rn_wgrad0_from_reductions) instead of a pass over dZ_0 and P."""
    return plan.inject == 0 and k <= 32 and OPT.algebraic_wgrad0


def alg0_forward_ok(plan, code, n, k, M, f16s=False):
    """The factored first layer (rn_g_chain_fwd_rr_alg0): bf16 register-resident chains, question injected at layer 0,
    whole waves per (question, i) -- f16s: on a padded j axis where n % 32 != 0 -- and the algebraic layer-0 weight gradient in
    the backward pass (nothing reads P).  M = B * n * n."""
    njp = padded_j(n)
    if njp is None or (njp != n and not f16s):
        return False
    return (rr_chain_ok(plan, code) and plan.inject == 0 and k <= 32 and (M // n * njp) % H.g_chain_rr_tile() == 0
            and OPT.rr_masks and OPT.algebraic_wgrad0
            and OPT.algebraic_fwd0)


def _tables(x, q, plan, g_b, w0T, inj_w, xdt, B, n, k, Q, G, coord=None):
    """Tables of the factored first layer (+ the question rows of an injected later layer): -> (Xp, Vc, Vq | None, inject).
    coord (2, n): x is the conv grid itself (kf = k - 2 columns), the coordinate tags are read from the table in the kernel."""
    dev = x.device
    Xp = torch.empty(B * n + 1, 64, dtype=xdt, device=dev)       # (+ the all-zero object row the padded-j chain reads for j >= n)
    if n % 32:                                                   # (only the padded-j chain reads the row: no fill launch otherwise)
        Xp[B * n].zero_()
    Vc = torch.empty(B * n, G, dtype=torch.float32, device=dev)
    if inj_w is None:
        H.pair_tables(x, q, w0T, g_b[0], Xp, Vc, B, n, k, Q, G, coord=coord)
        return Xp, Vc, None, 0
    inj = plan.inject
    H.pair_tables(x, None, w0T, g_b[0], Xp, Vc, B, n, k, 0, G, coord=coord)
    Gp = plan.widths[inj - 1]
    Vq = torch.empty(B, G, dtype=torch.float32, device=dev)
    # Vq[b, f] = b_inj[f] + sum_c q[b, c] W_inj[f, Gp + c]   (model.py:135-141: the question is the layer's trailing Q columns)
    H.gemm_f32(q, Q, 1, inj_w, 1, inj_w.shape[1], Vq, G, B, G, Q, bias=g_b[inj], b_off=Gp)
    return Xp, Vc, Vq, inj


def g_chain_forward(x, q, plan: LayerPlan, g_b, wfwd, code, keep_inputs=True, layer_hook=None, split=None, wfrag=None, stop_at=None,
                    w0T=None, inj_w=None, coord=None, lazy_xg=False):
    """K1 + K2 chain (+ K3).  Returns (inputs, H_L, xg): the list of layer INPUT buffers
    [A_0 .. A_{L-1}], the last activation H_L and -- when the fused chain ran -- the pair sum xg
    (else None).  layer_hook(l, A_l, H_out) is called after every layer (hook-compat path; forces
    the per-layer kernels)."""
    B, n, k = x.shape
    if coord is not None:
        k += coord.shape[0]
    Q = q.shape[1]
    M = B * n * n
    dt = H.torch_dtype(code)
    dev = x.device
    inj = plan.inject
    ld0 = plan.kpad[0]
    if split is not None:
        # "f16s": fp16 pair matrix + split fp16 weights through the fused chain; the bf16 pair matrix is only
        # needed by the backward pass (layer-0 wgrad)
        if layer_hook is not None or not (fused_chain_ok(plan, code, B, n) or inj_w is not None or w0T is not None):
            raise RuntimeError('precision "f16s" needs a fused chain (bf16-class storage, all g widths 256, question injected at '
                               'layer 0 with B*n*n a multiple of 128 -- or at layer 2 with n*n a multiple of 256 --, no forward '
                               'hooks); use "bf16" or "fp32" here')
        G, L, T = plan.widths[-1], plan.L, H.g_chain_tile()
        if w0T is not None and wfrag is not None and len(wfrag) == 2 and len(wfrag[0]) == L:
            # factored first layer in the f16s arithmetic: fp16 object rows + fp32 bias rows, no pair matrix
            R = H.g_chain_rr_tile()                    # (one pair-sum partial row per 256-row tile on these paths)
            Xp, Vc, Vq, inj_l = _tables(x, q, plan, g_b, w0T, inj_w, torch.float16, B, n, k, Q, G, coord)
            njp = n if inj_w is not None else padded_j(n)          # pair rows per (question, i) group: padded where n % 32 != 0
            Mp = B * n * njp
            masks = Hs = gate = None
            if keep_inputs:
                Hs = [torch.empty(Mp, G, dtype=_h_copy_dtype(plan, dt, Mp), device=dev) for l in range(L - 1)] + [None]
                masks = list(torch.empty(L, H.g_chain_rr_mask_bytes(Mp), dtype=torch.uint8, device=dev))
                if Hs[0].dtype in H.FP8_DTYPES and (n * njp) % 64 == 0 and OPT.gated_wgrad and OPT.gate_fwd:
                    # the operand of the last layer's weight-gradient gate job, written from the forward kernel's epilogue
                    gate = torch.empty(Mp, G, dtype=Hs[0].dtype, device=dev)
            if njp != n:
                part = torch.empty(Mp // R * 2, G, dtype=torch.float32, device=dev)      # two partial rows per tile (it may straddle questions)
                H.g_chain_fwd_rr_f16s_alg0(Xp, Vc, n, wfrag[0], wfrag[1][0], g_b, Hs, masks, part, Mp, G, njp=njp, gate=gate)
                xg = torch.empty(B, G, dtype=torch.float32, device=dev)
                H.pair_sum_tiles(part, xg, Mp, n * njp, G)
            else:
                part = torch.empty(M // R, G, dtype=torch.float32, device=dev)
                H.g_chain_fwd_rr_f16s_alg0(Xp, Vc, n, wfrag[0], wfrag[1][0], g_b, Hs, masks, part, M, G, Vq=Vq, inject=inj_l, gate=gate)
                xg = _pair_sum_of(part, B, (n * n) // R, G, lazy_xg)
            if Hs is None:
                return [None] * L, None, xg
            return [None] + Hs[:-1], RRMasks(masks, gate), xg
        P16 = torch.empty(M, ld0, dtype=torch.float16, device=dev)
        H.pair_build_fwd(x, q, P16, H.RN_F16, B, n, k, Q, ld0)
        P = None
        if keep_inputs and not alg0_wgrad_ok(plan, k):     # (the algebraic layer-0 weight gradient never reads P)
            P = torch.empty(M, ld0, dtype=dt, device=dev)
            H.pair_build_fwd(x, q, P, code, B, n, k, Q, ld0)
        if (wfrag is not None and len(wfrag) == 2 and len(wfrag[0]) == L and rr_chain_ok(plan, code) and M % H.g_chain_rr_tile() == 0
                and ((n * n) % 32 == 0 or keep_inputs) and OPT.rr_masks):
            # register-resident mapping: fp16 operand registers, hi + lo weight fragments; bf16 copies + lane masks for the
            # (shared, bf16) backward chain.  Waves that straddle two questions (n*n % 32 != 0, the 14x14 grid): the pair sum
            # comes from the stored H_3 instead of the in-lane partials (training only: inference has nowhere to store it)
            R = 32
            whole = (n * n) % R == 0
            masks = Hs = None
            if keep_inputs:
                # (e4m3 copies of H_0..2 here too when the layer-0 reduction is the 192-column one; a stored H_3 stays 16-bit)
                hdt = _h_copy_dtype(plan, dt, M) if ld0 == 192 else dt
                Hs = [torch.empty(M, G, dtype=hdt, device=dev) for l in range(L - 1)] + [None if whole else torch.empty(M, G, dtype=dt, device=dev)]
                masks = list(torch.empty(L, H.g_chain_rr_mask_bytes(M), dtype=torch.uint8, device=dev))
            part = torch.empty(M // R, G, dtype=torch.float32, device=dev) if whole else None
            H.g_chain_fwd_rr_f16s(P16, ld0, wfrag[0], wfrag[1][0], g_b, Hs, masks, ld0, part, M, G)
            xg = torch.empty(B, G, dtype=torch.float32, device=dev)
            if whole:
                H.pair_sum_fwd(part, G, xg, H.RN_F32, B, (n * n) // R, G)
            else:
                H.pair_sum_fwd(Hs[-1], G, xg, code, B, n * n, G)
            if Hs is None:
                return [P, None, None, None], None, xg
            return [P] + Hs[:-1], RRMasks(masks), xg
        whole = (n * n) % T == 0
        Hs = [torch.empty(M, G, dtype=dt, device=dev) if (keep_inputs or (l == L - 1 and not whole)) else None
              for l in range(L)]
        part = torch.empty(M // T, G, dtype=torch.float32, device=dev) if whole else None
        H.g_chain_fwd_f16s(P16, ld0, split[0], split[1], g_b, Hs, plan.kpad, part, M, G)
        xg = torch.empty(B, G, dtype=torch.float32, device=dev)
        if whole:
            H.pair_sum_fwd(part, G, xg, H.RN_F32, B, (n * n) // T, G)
        else:
            H.pair_sum_fwd(Hs[-1], G, xg, code, B, n * n, G)
        return [P] + Hs[:-1], Hs[-1], xg
    if w0T is not None and stop_at is None and layer_hook is None and wfrag is not None and len(wfrag) == plan.L:
        # factored first layer: two small tables instead of the pair matrix, K = 64 instead of 192 in layer 0
        G, L, R = plan.widths[-1], plan.L, H.g_chain_rr_tile()     # (one pair-sum partial row per 256-row tile)
        Xp, Vc, Vq, inj_l = _tables(x, q, plan, g_b, w0T, inj_w, dt, B, n, k, Q, G, coord)
        masks = Hs = None
        if keep_inputs:
            Hs = [torch.empty(M, G, dtype=_h_copy_dtype(plan, dt, M), device=dev) for l in range(L - 1)] + [None]
            masks = list(torch.empty(L, H.g_chain_rr_mask_bytes(M), dtype=torch.uint8, device=dev))
        part = torch.empty(M // R, G, dtype=torch.float32, device=dev)
        H.g_chain_fwd_rr_alg0(Xp, Vc, n, wfrag, g_b, Hs, masks, part, M, G, Vq=Vq, inject=inj_l)
        xg = _pair_sum_of(part, B, (n * n) // R, G, lazy_xg)
        if Hs is None:
            return [None] * L, None, xg
        return [None] + Hs[:-1], RRMasks(masks), xg
    if coord is not None:
        raise RuntimeError("internal: the coordinate table is only taken by the factored-first-layer paths")
    P = torch.empty(M, ld0, dtype=dt, device=dev)
    H.pair_build_fwd(x, q if inj == 0 else None, P, code, B, n, k, Q if inj == 0 else 0, ld0)
    if stop_at == 0:
        return [P], None, None
    if stop_at is None and layer_hook is None and fused_chain_ok(plan, code, B, n):
        G = plan.widths[-1]
        L = plan.L
        R = 32                                      # pair rows per wave of the register-resident chain
        if (wfrag is not None and len(wfrag) == L and rr_chain_ok(plan, code) and M % H.g_chain_rr_tile() == 0
                and (keep_inputs or (n * n) % R == 0)):
            whole = (n * n) % R == 0
            # training with whole waves per question: the last activation never leaves the chip -- its pair sum and
            # the ReLU gates of all layers (32 bytes per pair row and layer) do; rn_g_chain_bwd_rr consumes the gates
            masks = None
            Hs = None
            if keep_inputs:
                Hs = [torch.empty(M, G, dtype=dt, device=dev) for l in range(L)]
                if OPT.rr_masks:
                    masks = list(torch.empty(L, H.g_chain_rr_mask_bytes(M), dtype=torch.uint8, device=dev))
                    if whole:
                        Hs[-1] = None                       # (waves straddling questions: the pair sum needs the stored H_3)
            part = torch.empty(M // R, G, dtype=torch.float32, device=dev) if whole else None
            H.g_chain_fwd_rr(P, ld0, wfrag, g_b, Hs, masks, ld0, part, M, G)
            xg = torch.empty(B, G, dtype=torch.float32, device=dev)
            if whole:
                H.pair_sum_fwd(part, G, xg, H.RN_F32, B, (n * n) // R, G)
            else:
                H.pair_sum_fwd(Hs[-1], G, xg, code, B, n * n, G)
            if Hs is None:
                return [P, None, None, None], None, xg
            return [P] + Hs[:-1], (Hs[-1] if masks is None else RRMasks(masks)), xg
        T = H.g_chain_tile()
        whole = (n * n) % T == 0                    # whole tiles per question -> pair sum from the on-chip tiles
        # activations are stored only when the backward pass will need them (the last one also feeds the
        # stand-alone pair sum when tiles straddle questions)
        Hs = [torch.empty(M, G, dtype=dt, device=dev) if (keep_inputs or (l == L - 1 and not whole)) else None
              for l in range(L)]
        part = torch.empty(M // T, G, dtype=torch.float32, device=dev) if whole else None
        H.g_chain_fwd(P, ld0, wfwd, g_b, Hs, plan.kpad, part, code, M, G)
        xg = torch.empty(B, G, dtype=torch.float32, device=dev)
        if whole:
            H.pair_sum_fwd(part, G, xg, H.RN_F32, B, (n * n) // T, G)
        else:
            H.pair_sum_fwd(Hs[-1], G, xg, code, B, n * n, G)
        return [P] + Hs[:-1], Hs[-1], xg
    inputs = [P]
    cur = P
    for l in range(plan.L):
        N = plan.widths[l]
        nxt_wide = (l + 1 < plan.L) and (l + 1 == inj)
        ldh = plan.kpad[l + 1] if nxt_wide else N
        if nxt_wide and ldh > N + Q:
            out = torch.zeros(M, ldh, dtype=dt, device=dev)
        else:
            out = torch.empty(M, ldh, dtype=dt, device=dev)
        H.g_linear_fwd(cur, plan.kpad[l], wfwd[l], plan.kpad[l], g_b[l], out, ldh, code, M, N, plan.kpad[l])
        if nxt_wide:
            H.qst_broadcast(q, out, code, B, n, Q, N, ldh)
        if layer_hook is not None:
            layer_hook(l, cur, out)
        if stop_at is not None and l + 1 == stop_at:         # the caller wants the INPUT of layer stop_at only
            return inputs + [out], None, None
        if l + 1 < plan.L:
            inputs.append(out)
        if not keep_inputs and l >= 1:
            inputs[l] = None
        cur = out
    return inputs, cur, None


class PairSumPartials:
    """The forward chain's per-tile partial pair sums, not yet added up: (B * parts, G) fp32.  The f_phi launch adds them itself
    (rn_f_phi_fwd_from_partials) and writes the (B, G) sums into `.xg`."""

    def __init__(self, part, B, parts, G):
        self.part, self.B, self.parts, self.G = part, B, parts, G
        self.xg = torch.empty(B, G, dtype=torch.float32, device=part.device)


def _pair_sum_of(part, B, parts, G, lazy):
    if lazy and OPT.fused_pair_sum:
        return PairSumPartials(part, B, parts, G)
    xg = torch.empty(B, G, dtype=torch.float32, device=part.device)
    H.pair_sum_fwd(part, G, xg, H.RN_F32, B, parts, G)
    return xg


def f_phi_forward(xg, fw, fb, mask, wT=None, label=None, pre_bwd=False):
    """f_phi + log_softmax (model.py:155-162): fc1 -> relu -> fc2 -> dropout mask -> relu -> fc3 -> log_softmax,
    fp32.  Returns (f1, f2, log_probs), with `label` (int64 (B,)) also the mean NLL as a fourth element (same launch).
    xg: the (B, G) pair sums, or PairSumPartials (their summation then rides in the same launch; .xg holds them afterwards)."""
    lazy = xg if isinstance(xg, PairSumPartials) else None
    if lazy is not None:
        xg = lazy.xg
    B, G = xg.shape
    dev = xg.device
    F1, F2, A = fw[0].shape[0], fw[1].shape[0], fw[2].shape[0]
    f1 = torch.empty(B, F1, dtype=torch.float32, device=dev)
    f2 = torch.empty(B, F2, dtype=torch.float32, device=dev)
    out = torch.empty(B, A, dtype=torch.float32, device=dev)
    if lazy is not None and pre_bwd and label is not None and wT is not None:
        # the training step: the backward dz chain (for d loss = 1) rides in the same launch -> (.., loss, (dz workspace, dxg))
        loss = torch.empty((), dtype=torch.float32, device=dev)
        dxg = torch.empty(B, G, dtype=torch.float32, device=dev)
        ws = H.f_phi_fwd_bwd_from_partials(lazy.part, lazy.parts, xg, wT, fb, fw, mask, label, f1, f2, out, loss, dxg)
        return f1, f2, out, loss, (ws, dxg)
    if lazy is not None:
        loss = torch.empty((), dtype=torch.float32, device=dev) if label is not None else None
        H.f_phi_fwd_from_partials(lazy.part, lazy.parts, xg, wT if wT is not None else fw, fb, mask, label, f1, f2, out, loss,
                                  transposed=wT is not None)
        return (f1, f2, out, loss) if label is not None else (f1, f2, out)
    if label is not None:
        loss = torch.empty((), dtype=torch.float32, device=dev)
        H.f_phi_fwd_nll(xg, wT if wT is not None else fw, fb, mask, label, f1, f2, out, loss, transposed=wT is not None)
        return f1, f2, out, loss
    if wT is not None:
        H.f_phi_fwd(xg, wT, fb, mask, f1, f2, out, transposed=True)      # one launch (rn_small.hip), coalesced weight reads
    else:
        H.f_phi_fwd(xg, fw, fb, mask, f1, f2, out)
    return f1, f2, out


class RelationalFunction(torch.autograd.Function):
    """(x, q, dropout_mask | None, plan, packed, precision, label | None, coord | None, g_w.., g_b.., f_w.., f_b..) -> log-probs
    (B, A), or with `label` (int64 (B,)) -> (log-probs, mean NLL): the loss of train.py:41 rides in the f_phi launches.
    coord (2, n) fp32: x is the conv grid viewed (B, n, k - 2) -- the coordinate tags of model.py:195-201 are applied inside
    the kernels and the input gradient comes back in the grid's own layout (no concatenation, no slicing)."""

    @staticmethod
    def forward(ctx, x, q, mask, plan, packed, precision, label, coord, *params):
        ctx.set_materialize_grads(False)
        L = plan.L
        g_w, g_b = params[0:L], params[L:2 * L]
        f_w, f_b = params[2 * L:2 * L + 3], params[2 * L + 3:2 * L + 6]
        H._dev(x, "x")
        H._dev(q, "qst")
        code = H.dtype_code(precision)
        x = x.float() if x.dtype != torch.float32 else x
        q = q.float().contiguous() if (q.dtype != torch.float32 or not q.is_contiguous()) else q
        B, n, k = x.shape
        if coord is not None:
            k += coord.shape[0]
        Q = q.shape[1]
        M = B * n * n
        dev = x.device
        f16s = precision == "f16s"
        need_grad = any(ctx.needs_input_grad)
        # exactly g_chain_forward's condition for the register-resident branches (they consume only the fragment-major
        # images): whole waves per question, or -- waves straddling questions -- a stored H_3 (training only)
        rr_only = (rr_chain_ok(plan, code) and M % H.g_chain_rr_tile() == 0 and OPT.rr_masks
                   and ((n * n) % 32 == 0 or need_grad))
        alg_fwd = alg0_forward_ok(plan, code, n, k, M, f16s=f16s)
        njp = padded_j(n) if (alg_fwd and plan.inject == 0) else n           # (padded j axis: n % 32 != 0, f16s)
        rr_only = rr_only or alg_fwd
        inj_fwd = inj_chain_ok(plan, code, n, k, M)        # question injected at layer 2: same chains, per-question bias row
        if inj_fwd:
            rr_only = alg_fwd = True
        if coord is not None and not alg_fwd:
            raise RuntimeError("internal: a coordinate table was passed but the factored-first-layer path does not apply (grid_fast_path)")
        wfwd, wbwd = packed.get(plan, g_w, code, split=f16s, bwd_images=need_grad, rr_only=rr_only, f_w=f_w,
                                alg0_k=k if alg_fwd else 0, inj=plan.inject if inj_fwd else 0)
        gb = [b.detach().contiguous() for b in g_b]
        inj_w = None
        if inj_fwd:
            inj_w = g_w[plan.inject].detach()
            inj_w = inj_w if inj_w.is_contiguous() else inj_w.contiguous()
        inputs, HL, xg = g_chain_forward(x, q, plan, gb, wfwd, code, keep_inputs=need_grad,
                                         split=(packed.hi, packed.lo) if f16s else None,
                                         wfrag=(packed.frag_hi, packed.frag_lo) if f16s else packed.frag,
                                         w0T=packed.w0T if alg_fwd else None, inj_w=inj_w, coord=coord, lazy_xg=True)
        G = plan.widths[-1]
        if COPY_HEALTH_PROBE is not None and need_grad and isinstance(HL, RRMasks):
            for l in range(1, L):                              # inputs[l] = the copy of H_{l-1}
                if inputs[l] is not None and inputs[l].dtype in H.FP8_DTYPES:
                    COPY_HEALTH_PROBE.append((l - 1, H.fp8_copy_health(HL.masks[l - 1], inputs[l], inputs[l].numel() // G)))
        if xg is None:
            xg = torch.empty(B, G, dtype=torch.float32, device=dev)
            H.pair_sum_fwd(HL, G, xg, code, B, n * n, G)
        fw = [w.detach().contiguous() for w in f_w]
        fb = [b.detach().contiguous() for b in f_b]
        F1, F2, A = fw[0].shape[0], fw[1].shape[0], fw[2].shape[0]
        if mask is not None:
            mask = mask.float().contiguous()
        loss = None
        ctx.fphi_pre = None
        if label is not None:
            label = label.long().contiguous()
            r = f_phi_forward(xg, fw, fb, mask, wT=packed.fT, label=label,
                              pre_bwd=need_grad and OPT.fphi_fused_bwd and isinstance(xg, PairSumPartials))
            f1, f2, out, loss = r[:4]
            if len(r) == 5:
                ctx.fphi_pre = r[4]
        else:
            f1, f2, out = f_phi_forward(xg, fw, fb, mask, wT=packed.fT)
        if isinstance(xg, PairSumPartials):
            xg = xg.xg
        ctx.label = label
        if need_grad:
            ctx.plan, ctx.code, ctx.dims = plan, code, (B, n, k, Q, M, G, F1, F2, A)
            ctx.njp = njp if isinstance(HL, RRMasks) else n
            ctx.inj_path = inj_fwd
            ctx.coord = coord
            ctx.inputs, ctx.HL, ctx.wbwd = inputs, HL, wbwd
            ctx.fragT = list(packed.fragT)
            ctx.g_w = [w.detach() for w in g_w]
            ctx.param_refs = list(g_w) + list(g_b)
            ctx.f_param_refs = list(f_w) + list(f_b)
            ctx.fw = fw
            ctx.mask = mask
            ctx.save_for_backward(x, q, xg, f1, f2, out)
        return out if label is None else (out, loss)

    @staticmethod
    def backward(ctx, gout, gloss=None):
        plan, code = ctx.plan, ctx.code
        B, n, k, Q, M, G, F1, F2, A = ctx.dims
        x, q, xg, f1, f2, out = ctx.saved_tensors
        dev = x.device
        L = plan.L
        f32 = dict(dtype=torch.float32, device=dev)
        label = ctx.label
        if gout is not None and gloss is not None:          # both outputs were used: fold the loss term into the log-prob gradient
            gout = gout.float().clone()
            gout[torch.arange(B, device=dev), label] -= gloss.float() / B
            gloss = None
        if gout is None and gloss is None:
            gout = torch.zeros(B, A, **f32)
        if gout is not None:
            gout = gout.float().contiguous()
        else:
            gloss = gloss.float().contiguous()
        fw = ctx.fw
        # ---- f_phi backward (fp32): two launches (dz chain incl. log_softmax; all weight / bias gradients)
        fp = ctx.f_param_refs                                # (f_fc1..3 weights, then biases)
        dW3 = grad_out(fp[2], (A, F2)); db3 = grad_out(fp[5], (A,))
        dW2 = grad_out(fp[1], (F2, F1)); db2 = grad_out(fp[4], (F2,))
        dW1 = grad_out(fp[0], (F1, G)); db1 = grad_out(fp[3], (F1,))
        if gout is None and ctx.fphi_pre is not None and is_unit_loss_grad(gloss):
            # the forward launch already ran the dz chain for d loss = 1: the parameter gradients are all that is left
            ws_, dxg = ctx.fphi_pre
            H.f_phi_bwd_grads(ws_, xg, f1, f2, (dW1, dW2, dW3), (db1, db2, db3))
        elif gout is None:
            dxg = torch.empty(B, G, **f32)
            H.f_phi_bwd_nll(gloss, label, out, f2, f1, xg, fw, ctx.mask, (dW1, dW2, dW3), (db1, db2, db3), dxg)
        else:
            dxg = torch.empty(B, G, **f32)
            H.f_phi_bwd(gout, out, f2, f1, xg, fw, ctx.mask, (dW1, dW2, dW3), (db1, db2, db3), dxg)
        # ---- g_theta backward
        dt = H.torch_dtype(code)
        inputs, wbwd, g_w = ctx.inputs, ctx.wbwd, ctx.g_w
        fused_bwd = fused_chain_ok(plan, code, B, n) and L >= 2 and OPT.fused_bwd
        rr_bwd = isinstance(ctx.HL, RRMasks)       # the register-resident chains: H_0..2 / dZ of layers 1..3 are row-blocked images
        njp = ctx.njp                              # pair rows per (question, i) group: > n on the padded j axis (rr chains only)
        Mc = B * n * njp                           # ... and the pair rows the chains / weight gradients work on
        gate_img = None
        if rr_bwd:
            # register-resident backward chain on the forward kernel's gates (one launch, no activation is re-read)
            fused_bwd = True
            # the last layer's gradient dZ_{L-1} = gate x dxg[question] is never stored: its only reader besides the chain
            # itself, the layer's wgrad, rebuilds it from the masks (rn_g_wgrad_blocked) -- 134 MB less written
            # by the chain and 134 MB less read by the wgrad at the headline shape
            gated_mask = gate_img = None
            # (needs the e4m3 H_2 image: the gate job runs on the fp8 matrix pipe; with 16-bit copies dZ_3 is stored)
            red_parts = None
            if ((n * njp) % 64 == 0 and inputs[L - 1].dtype in H.FP8_DTYPES and OPT.gated_wgrad):
                gated_mask = ctx.HL.masks[L - 1]
                gate_img = ctx.HL.gate                     # (the f16s forward chain has already written the gate's image)
                # layer 0's gradient is read by the pair-axis reductions only: they are formed inside the chain and dZ_0
                # (134 MB written + read back at the headline shape) never exists -- rn_g_chain_bwd_rr_red
                tpu = H.g_chain_bwd_rr_red_tpu(Mc, n) if (OPT.chain_reduce and njp == n and L == 4 and k <= 32
                                                          and (alg0_wgrad_ok(plan, k) or bool(ctx.inj_path))) else 0
                if tpu > 0:
                    dZs = [None] + list(torch.empty(L - 2, Mc, G, dtype=dt, device=dev)) + [None]
                    red_parts = (torch.empty(Mc // 256 // tpu, 32, G, **f32), torch.empty(Mc // 16, G, **f32), (n // 8) // tpu)
                    H.g_chain_bwd_rr_red(dxg, ctx.HL.masks, ctx.fragT, dZs, Mc, n, G, red_parts[0], red_parts[1], tpu)
                else:
                    dZs = [None] + list(torch.empty(L - 1, Mc, G, dtype=dt, device=dev))
            else:
                dZs = list(torch.empty(L, Mc, G, dtype=dt, device=dev))            # dZs[s] belongs to layer L-1-s
            if red_parts is None:
                H.g_chain_bwd_rr(dxg, ctx.HL.masks, ctx.fragT, dZs, Mc, n * njp, G)
            dZ_of = {L - 1 - s: dZs[s] for s in range(L)}
        elif fused_bwd:
            # one launch: dZ_L = dxg * (H_L > 0), then dZ_{l-1} = (dZ_l @ W_l) * (H_{l-1} > 0) for every layer
            dZs = [torch.empty(M, G, dtype=dt, device=dev) for _ in range(L)]      # dZs[s] belongs to layer L-1-s
            H.g_chain_bwd(ctx.HL, dxg, [wbwd[L - 1 - s] for s in range(L - 1)], [inputs[L - 1 - s] for s in range(L - 1)],
                          dZs, code, M, n * n, G)
            dZ_of = {L - 1 - s: dZs[s] for s in range(L)}
        else:
            dZ = torch.empty(M, G, dtype=dt, device=dev)
            H.pair_sum_bwd(dxg, ctx.HL, G, dZ, G, code, B, n * n, G)
        if not isinstance(ctx.HL, RRMasks):
            gated_mask = None
        ctx.HL = None
        gW, gB = [None] * L, [None] * L
        dq = None
        dx = None
        inj = bool(ctx.inj_path)        # the chains ran with the question injected at layer plan.inject > 0 as a bias row


        def _pair_reduce(dz0, Rj, Ri, Rq, N_):
            if rr_bwd and red_parts is not None:                       # the chain has already reduced: add its partials up
                H.pair_reduce_parts(red_parts[0], red_parts[1], Rj, Ri, Rq, B, n, N_, red_parts[2])
            else:
                H.pair_reduce_bwd(dz0, N_, Rj, Ri, Rq, code, B, n, N_, njp=njp)

        def _wgrad(l, dz, a_l):                                        # row-major operands: the general kernel
            N_, kt_, kp_ = plan.widths[l], plan.ktrue[l], plan.kpad[l]
            gW[l] = grad_out(ctx.param_refs[l], (N_, kt_))
            gB[l] = grad_out(ctx.param_refs[L + l], (N_,))
            H.g_linear_bwd_wgrad(dz, N_, a_l, kp_, gW[l], gB[l], code, M, N_, kp_, kt_)

        def _wgrads_blocked(dz_all, a_all):
            """Layers 1..L-1 on the register-resident chains' row-blocked images: ONE launch + one reduction launch
            (rn_g_wgrad_blocked).  The injected layer (input [H_{l-1} | q]): dW = [dZ^T H_{l-1} | Rq^T q] -- the H part is an
            ordinary job; Rq (per-question column sums of dZ) = the kernel's bias-gradient partials when no row split
            straddles two questions, dq and the question columns of dW follow at once, on this stream."""
            order = list(range(1, L))
            if inj:                                                # (its dq is waited for by the main stream: first)
                order.remove(plan.inject)
                order.insert(0, plan.inject)
            jobs, tmp = [], None
            for l in order:
                N_, kt_ = plan.widths[l], plan.ktrue[l]
                gW[l] = grad_out(ctx.param_refs[l], (N_, kt_))
                gB[l] = grad_out(ctx.param_refs[L + l], (N_,))
                # the last layer without a stored gradient: its gate as an e4m3 {0, 1} image, scaled by dxg per question in the kernel
                dz_l = dz_all[l] if dz_all[l] is not None else (gate_img if gate_img is not None else H.relu_gate_image(gated_mask, Mc))
                if inj and l == plan.inject:
                    tmp = torch.empty(N_, plan.widths[l - 1], **f32)
                    jobs.append((dz_l, a_all[l], tmp, gB[l]))
                else:
                    jobs.append((dz_l, a_all[l], gW[l], gB[l]))
            ws_, parts = H.g_wgrad_blocked(jobs, Mc, dxg=dxg, rows_per_question=n * njp, aligned=bool(rq_splits))
            if tmp is not None:
                l = plan.inject
                N_, kt_, gp = plan.widths[l], plan.ktrue[l], plan.widths[l - 1]
                gW[l][:, :gp].copy_(tmp)
                if rq_splits:
                    wl_ = g_w[l] if g_w[l].is_contiguous() else g_w[l].contiguous()
                    rq_ = parts[0].view(B, (rq_splits // B) * 4, N_).sum(1)
                    dq_ = torch.empty(B, Q, **f32)
                    H.gemm_f32(rq_, N_, 1, wl_, kt_, 1, dq_, Q, B, Q, N_, b_off=kt_ - Q)          # Rq @ W[:, -Q:]
                    H.gemm_f32(rq_, 1, N_, q, Q, 1, gW[l], kt_, N_, Q, B, c_off=kt_ - Q)          # gW[l][:, G_prev:] = Rq^T q
                    inj_out["dq"], inj_out["keep"] = dq_, [ws_, rq_, wl_, tmp]
        # The weight gradients are not needed by anything upstream: with the fused chain all dZ_l exist now, so
        # the L wgrad launches go to a side stream and overlap the rest of this backward AND the conv / LSTM
        # backward that autograd runs next (small kernels that leave the chip mostly empty).  The main stream
        # re-joins at the end of the backward pass (engine callback).  Only when every parameter's .grad is
        # None (assign, not accumulate: autograd then launches no kernel on these tensors before the join).
        overlap = (fused_bwd and OPT.wgrad_overlap
                   and all(_assign_only(p) for p in ctx.param_refs))
        # Layer 0 reads the pair matrix P = [x_j | x_i | q]: its weight gradient dZ_0^T P factors through the pair
        # reductions the input gradient needs anyway -- dW_0 = [Rj^T X | Ri^T X | Rq^T Q], db_0 = sum_b Rq -- three tiny
        # products on (B*n)-row matrices instead of a 235 MB pass over dZ_0 and P (and with fp32 x instead of P's
        # rounded copy).  RN_NO_ALGEBRAIC_WGRAD0=1 keeps the kernel.
        alg0 = alg0_wgrad_ok(plan, k) or (inj and k <= 32)
        # question injected at layer > 0: its per-question sums Rq from the wgrad kernel's per-split column sums when no split
        # straddles two questions (64 splits: B | 64), else from the pair-reduction kernel
        rq_splits, inj_out = 0, {}
        if inj and rr_bwd and OPT.rq_from_wgrad:
            z_ = H.wgrad_blocked_splits(M, n * n, L - 1, aligned=True)
            if z_ > 0 and z_ % B == 0 and (M // 64) % z_ == 0 and (n * n) % (M // z_) == 0:
                rq_splits = z_
        # RN_WGRAD_LATE=1 / 2 starts the (HBM-bound) wgrad stream only after the pair reduction / after dx, dq: measured
        # slower (1.102 / 1.137 vs 1.092 ms; all wgrads serially at the very end of the backward pass: 1.165) -- the window after the backward chain runs at the HBM roofline (~4 TB/s over
        # wgrad + pair reduction, tools/step_timeline.py) wherever the wgrads are put, and later they slow the conv backward
        wgrad_late = OPT.wgrad_late if (alg0 and not inj) else 0
        if overlap:
            main, side = torch.cuda.current_stream(), _side_stream(dev)
            keep = [list(dZs), list(inputs), gated_mask, gate_img, dxg]      # keep operands alive until the join
            dz_all = dict(dZ_of)

            def _launch_wgrads():
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    if rr_bwd:
                        _wgrads_blocked(dz_all, inputs_all)
                        if rq_splits:
                            inj_out["event"] = side.record_event()
                        if not alg0:
                            _wgrad(0, dz_all[0], inputs_all[0])
                    else:
                        for l in range(L):
                            if l == 0 and alg0:
                                continue                           # layer 0: from the pair reductions, below
                            _wgrad(l, dz_all[l], inputs_all[l])
            inputs_all = list(inputs)
            # (capturing this fork BEHIND the main stream's next launch -- same dependencies, the trick that removed the launch gaps of
            # the conv-stack backward -- made the step 17 % slower here: the replayed graph's queue mapping is not ours to steer)
            if not wgrad_late:
                _launch_wgrads()

            def _join():
                torch.cuda.current_stream().wait_stream(side)
                keep.clear()
            torch.autograd.Variable._execution_engine.queue_callback(_join)
        for l in reversed(range(L)):
            N = plan.widths[l]
            A_l = inputs[l]
            kt, kp = plan.ktrue[l], plan.kpad[l]
            if fused_bwd:
                dZ = dZ_of.pop(l)
            if not overlap and rr_bwd and l == L - 1:
                _wgrads_blocked({**dZ_of, l: dZ}, inputs)             # layers 1..L-1 at once (their dZ all exist)
            if not overlap and not (l == 0 and alg0) and not (rr_bwd and l > 0):
                _wgrad(l, dZ, A_l)
            wl = g_w[l] if g_w[l].is_contiguous() else g_w[l].contiguous()
            fused_tail = (l == 0 and (plan.inject == 0 or inj) and k <= 32
                          and OPT.fused_pair_tail)
            if l == plan.inject and l > 0 and rq_splits:
                pass                                               # Rq, dq and the question columns of dW: done with the layer's wgrad
            elif l == plan.inject:
                Rq = torch.empty(B, N, **f32)
                if l == 0:
                    Rj = torch.empty(B * n, N, **f32); Ri = torch.empty(B * n, N, **f32)
                    _pair_reduce(dZ, Rj, Ri, Rq, N)
                elif rr_bwd:
                    H.blocked_question_sums(dZ, Rq, M, n * n)      # (this layer's dZ is a row-blocked image)
                else:
                    H.pair_reduce_bwd(dZ, N, None, None, Rq, code, B, n, N)
                dq = torch.empty(B, Q, **f32)
                if not fused_tail:
                    H.gemm_f32(Rq, N, 1, wl, kt, 1, dq, Q, B, Q, N, b_off=kt - Q)   # Rq @ W[:, -Q:]
                if inj and l > 0:
                    def _wgrad_question(Rq_=Rq, l_=l, kt_=kt, N_=N):               # gW[l][:, G_prev:] = Rq^T q
                        H.gemm_f32(Rq_, 1, N_, q, Q, 1, gW[l_], kt_, N_, Q, B, c_off=kt_ - Q)
                    if overlap:
                        side.wait_stream(main)
                        with torch.cuda.stream(side):
                            _wgrad_question()
                        keep.append([Rq, q])
                    else:
                        _wgrad_question()
            elif l == 0:
                Rj = torch.empty(B * n, N, **f32); Ri = torch.empty(B * n, N, **f32)
                Rq = torch.empty(B, N, **f32) if alg0 else None                   # (all-pairs sums: the layer's bias gradient)
                _pair_reduce(dZ, Rj, Ri, Rq, N)
            if l == 0 and overlap and wgrad_late == 1:
                _launch_wgrads()
            if l == 0 and alg0:
                def _wgrad0():
                    gW[0] = grad_out(ctx.param_refs[0], (N, kt))
                    gB[0] = grad_out(ctx.param_refs[L], (N,))
                    H.wgrad0_from_reductions(Rj, Ri, Rq, x, q if plan.inject == 0 else None, gW[0], gB[0], coord=ctx.coord)
                # (measured: on the conv weight-gradient stream instead -3.5 %, on the main stream behind dx / dq -1 %)
                if overlap and OPT.wgrad0_stream == 1:
                    pass                                           # (main stream, behind dx / dq: below)
                elif overlap:                                      # off the critical path: onto the wgrad stream (or one of its own)
                    s0 = side if OPT.wgrad0_stream == 0 else _side_stream(dev, 2)
                    s0.wait_stream(main)
                    with torch.cuda.stream(s0):
                        _wgrad0()
                    if s0 is not side:
                        side.wait_stream(s0)                       # (the join at the end of the backward pass waits for `side`)
                    # x and q too: they are alive only through this node's saved tensors, which autograd releases as soon as
                    # backward() returns -- the caching allocator would hand their blocks to the conv / LSTM backward that the
                    # main stream runs next while this side-stream kernel still reads them (seen as a wrong dW_0 on a busy GPU)
                    keep.append([Rj, Ri, Rq, x, q, ctx.coord])
                else:
                    _wgrad0()
            if l == 0 and fused_tail:
                if ctx.coord is not None:
                    # the gradient goes straight into the conv grid's layout (B, k - 2, n): the two coordinate columns carry
                    # none (model.py:216) and autograd's view-backward hands the conv stack a contiguous tensor
                    dx = torch.empty(B, x.shape[2], n, **f32).permute(0, 2, 1)
                else:
                    dx = torch.empty(B, n, k, **f32)
                if plan.inject == 0:
                    H.pair_dx_dq(Rj, Ri, Rq, wl, dx, dq, B, n, k, Q, N)                    # dx and dq in one launch
                else:
                    H.pair_dx_dq(Rj, Ri, None, wl, dx, None, B, n, k, 0, N)                # (dq came from the injected layer)
                if overlap and wgrad_late == 2:
                    _launch_wgrads()
            elif l == 0:
                dx = torch.empty(B, n, k, **f32)
                H.gemm_f32(Rj, N, 1, wl, kt, 1, dx, k, B * n, k, N)                        # Rj @ W0[:, 0:k]
                H.gemm_f32(Ri, N, 1, wl, kt, 1, dx, k, B * n, k, N, b_off=k, flags=H.RN_ACCUMULATE)   # + Ri @ W0[:, k:2k]
            elif not fused_bwd:
                gp = plan.widths[l - 1]
                dZp = torch.empty(M, gp, dtype=dt, device=dev)
                H.g_linear_bwd_dgrad(dZ, N, wbwd[l], N, A_l, kp, dZp, gp, code, M, N, gp)
                dZ = dZp
            if l == 0 and alg0 and overlap and OPT.wgrad0_stream == 1:
                _wgrad0()                                          # main stream, behind dx / dq
            inputs[l] = None
        ctx.inputs = None
        if rq_splits:
            dq = inj_out["dq"]
            if "event" in inj_out:                                 # produced on the wgrad stream
                torch.cuda.current_stream().wait_event(inj_out["event"])
                dq.record_stream(torch.cuda.current_stream())
                keep.append(inj_out["keep"])
        grads = [dx if ctx.needs_input_grad[0] else None, dq if ctx.needs_input_grad[1] else None, None, None, None, None, None, None]
        grads += gW + gB + [dW1, dW2, dW3, db1, db2, db3]
        return tuple(grads)


def relational_forward(x, q, mask, plan, packed, precision, g_w, g_b, f_w, f_b, label=None, coord=None):
    """-> log-probs, or (log-probs, mean NLL) when `label` is given."""
    if precision not in PRECISIONS:
        raise ValueError("precision must be one of %r" % (PRECISIONS,))
    return RelationalFunction.apply(x, q, mask, plan, packed, precision, label, coord, *g_w, *g_b, *f_w, *f_b)


def grid_path_ok(plan: LayerPlan, precision, B, n, k):
    """Shapes / modes whose kernels take the conv grid + the coordinate table directly (the factored-first-layer paths)."""
    if precision not in ("bf16", "f16s") or not OPT.grid_fast:
        return False
    code, M = H.RN_BF16, B * n * n
    if not OPT.fused_pair_tail:                   # (the un-fused tail writes dx as (B, n, k): not the grid view's shape)
        return False
    if inj_chain_ok(plan, code, n, k, M):
        return True
    return alg0_forward_ok(plan, code, n, k, M, f16s=precision == "f16s") and alg0_wgrad_ok(plan, k)


def _direct_conv_ok(inp, conv_w, stride, padding):
    return (OPT.direct_conv and inp.dtype == torch.float32 and tuple(conv_w.shape[2:]) == (3, 3)
            and tuple(stride) == (2, 2) and tuple(padding) == (1, 1) and conv_w.shape[0] == 24 and conv_w.shape[1] in (3, 24)
            and inp.shape[2] % 2 == 0 and inp.shape[3] % 2 == 0)


def _assign_only(p):
    """True when autograd will only ASSIGN the gradient it is handed for leaf `p` -- no accumulation kernel, no hook --
    so that a gradient still being written on a side stream is not touched before the end-of-backward join.  With an
    existing .grad (FlatGradBucket's accumulate mode, zero_grad(set_to_none=False), micro-batch accumulation) or any
    hook, AccumulateGrad runs `grad += dw` / the hook on the main stream: the producer must then run there too."""
    if p is None or p.grad is not None:
        return False
    if getattr(p, "_backward_hooks", None) or getattr(p, "_post_accumulate_grad_hooks", None):
        return False
    return True


class ConvBNReLUFunction(torch.autograd.Function):
    """relu(batch_norm(conv2d(x))) of the reference's ConvInputModel block (model.py:22-35): the convolution is
    MIOpen's (aten.convolution, run WITHOUT its bias), batch norm + ReLU and their backward are the fused
    two-pass kernels of rn_convnorm.hip.  A bias in front of a batch norm only shifts the batch mean -- it is
    added to the running mean and its gradient is identically zero (returned as zeros)."""

    @staticmethod
    def forward(ctx, inp, conv_w, conv_b, gamma, beta, running_mean, running_var, num_batches, training, momentum, eps, stride, padding):
        H._dev(inp, "img")
        direct = _direct_conv_ok(inp, conv_w, stride, padding)
        if direct:                                   # rn_conv.hip: direct 3x3 / stride-2 kernel
            inp = inp.contiguous()
            wc = conv_w.detach().contiguous()
            x = torch.empty(inp.shape[0], conv_w.shape[0], inp.shape[2] // 2, inp.shape[3] // 2, dtype=torch.float32, device=inp.device)
            H.conv3x3s2_fwd(inp, wc, x)
        else:
            x = torch.ops.aten.convolution(inp, conv_w, None, stride, padding, (1, 1), False, (0, 0), 1)
            x = x.contiguous()
        ctx.direct = direct
        Cc = x.shape[1]
        y = torch.empty_like(x)
        f32 = dict(dtype=torch.float32, device=x.device)
        g, bt = gamma.detach().contiguous(), beta.detach().contiguous()
        if training:
            mean = torch.empty(Cc, **f32); invstd = torch.empty(Cc, **f32)
            H.bn_relu_fwd(x, y, g, bt, conv_b.detach() if conv_b is not None else None, running_mean, running_var, num_batches,
                          mean, invstd, eps, momentum)
        else:
            mean = running_mean - (conv_b.detach() if conv_b is not None else 0)
            invstd = torch.rsqrt(running_var + eps)
            H.bn_relu_apply(x, y, g, bt, mean.contiguous(), invstd.contiguous())
        ctx.trained = training
        if training and any(ctx.needs_input_grad):
            ctx.save_for_backward(inp, conv_w, x, g, bt, mean, invstd)
            ctx.conv_args = (stride, padding)
            ctx.has_bias = conv_b is not None
            ctx.w_ref = conv_w                   # the leaf itself: backward looks at .grad / hooks (see _assign_only)
            ctx.leaf_refs = (conv_b, gamma, beta)
        return y

    @staticmethod
    def backward(ctx, dy):
        if not ctx.trained:         # ConvInputModel.forward keeps evaluation-mode graphs on the stock ops
            raise RuntimeError("ConvBNReLUFunction: backward through evaluation-mode batch norm is not implemented")
        inp, conv_w, x, g, bt, mean, invstd = ctx.saved_tensors
        stride, padding = ctx.conv_args
        dy = dy.contiguous()
        b_ref, gamma_ref, beta_ref = ctx.leaf_refs
        dgamma = grad_out(gamma_ref); dbeta = grad_out(beta_ref)
        if ctx.direct and not ctx.needs_input_grad[0] and OPT.direct_conv_wgrad and OPT.bn_wgrad_fused:
            # first block (the image needs no gradient) = the END of the backward pass: the conv output gradient is formed inside
            # the weight-gradient kernel and never written
            db = grad_out(b_ref) if ctx.has_bias else None
            dw = grad_out(ctx.w_ref)
            H.bn_relu_bwd_conv_wgrad(dy, x, inp, g, bt, mean, invstd, dgamma, dbeta, dw, zero_out=db)
            return None, dw, db, dgamma, dbeta, None, None, None, None, None, None, None, None
        dx = torch.empty_like(x)
        # the conv-bias gradient (identically zero) is a tensor of its own, zeroed by the same launch: a shared zero vector would be
        # CLONED by autograd for every leaf it is handed to -- one memcpy node per layer on the critical path of the captured step
        db = grad_out(b_ref) if ctx.has_bias else None
        H.bn_relu_bwd(dy, x, dx, g, bt, mean, invstd, dgamma, dbeta, zero_out=db)
        conv_bwd = lambda mask: torch.ops.aten.convolution_backward(dx, inp, conv_w, None, stride, padding, (1, 1), False, (0, 0), 1, mask)
        if ctx.needs_input_grad[0] and OPT.wgrad_overlap and _assign_only(ctx.w_ref):
            # only the input gradient is on the dependency chain of the backward pass: the weight gradient (MIOpen's
            # wrw kernel plus its layout transposes, ~half of the conv backward) goes to the side stream and overlaps the
            # next layers' backward; the main stream re-joins at the end of the backward pass
            # Captured order and stream matter here (tools/step_timeline.py, same box): the input gradient is launched FIRST and the
            # weight gradient forks off the event recorded before it -- with the fork in front, every bwd_data launch of the
            # captured step started 10-18 us late -- and the weight gradients share the g_theta weight-gradient stream instead of a
            # stream of their own: the ROCm graph executor maps the capture's streams onto 4 hardware queues, and a fourth side
            # stream ended up behind another one's kernels (the LSTM backward then started 100 us late).  Together -2.5 % on the step.
            main, side = torch.cuda.current_stream(), _side_stream(dx.device, 0)
            ev = main.record_event()
            if ctx.direct and inp.shape[1] == 24:
                din = torch.empty_like(inp)
                H.conv3x3s2_bwd_data(dx, conv_w.detach().contiguous(), din)
            else:
                din = conv_bwd([True, False, False])[0]
            side.wait_event(ev)
            with torch.cuda.stream(side):
                if ctx.direct and OPT.direct_conv_wgrad:
                    dw = grad_out(ctx.w_ref)
                    H.conv3x3s2_bwd_weight(inp, dx, dw)                # fp32 matrix pipe, no layout transposes
                else:
                    dw = conv_bwd([False, True, False])[1]
            keep = [dx, inp]
            for t in keep:
                t.record_stream(side)
            dw.record_stream(main)               # allocated on the side stream, handed to autograd on the main one

            def _join():
                torch.cuda.current_stream().wait_stream(side)
                keep.clear()
            torch.autograd.Variable._execution_engine.queue_callback(_join)
        elif ctx.direct and not ctx.needs_input_grad[0] and OPT.direct_conv_wgrad:
            din = None                                                 # first layer (the image needs no gradient): the END of
            dw = grad_out(ctx.w_ref)                                   # the backward pass, nothing left to overlap with
            H.conv3x3s2_bwd_weight(inp, dx, dw)
        else:
            din, dw, _ = conv_bwd([ctx.needs_input_grad[0], True, False])
        return din, dw, db, dgamma, dbeta, None, None, None, None, None, None, None, None


class QuestionLSTMFunction(torch.autograd.Function):
    """Embedding + 1-layer LSTM -> final hidden state (reference model.py:51-58) through rn_lstm.hip: the recurrence is
    one launch per direction; the parameter gradients are three small matrix products over the saved per-step
    matrices (rocBLAS) plus a deterministic embedding gather-add."""

    @staticmethod
    def forward(ctx, idx, emb, W_ih, W_hh, b_ih, b_hh):
        H._dev(idx, "question")
        idx = idx.long().contiguous()
        B, T = idx.shape
        E, Hh = emb.shape[1], W_hh.shape[1]
        f32 = dict(dtype=torch.float32, device=idx.device)
        ws = [t.detach().contiguous() for t in (emb, W_ih, W_hh, b_ih, b_hh)]
        train = any(ctx.needs_input_grad)
        if train:
            xs = torch.empty(T, B, E, **f32); gates = torch.empty(T, B, 4 * Hh, **f32)
            cs = torch.empty(T, B, Hh, **f32); hs = torch.empty(T + 1, B, Hh, **f32)
            H.lstm_fwd(idx, *ws, xs, gates, cs, hs)
            ctx.save_for_backward(idx, xs, gates, cs, hs, ws[1], ws[2])
            ctx.vocab = emb.shape[0]
            ctx.leaf_refs = (emb, W_ih, W_hh, b_ih, b_hh)
            return hs[T].clone()
        hn = torch.empty(B, Hh, **f32)
        H.lstm_fwd(idx, *ws, None, None, None, hn)
        return hn

    @staticmethod
    def backward(ctx, dhn):
        idx, xs, gates, cs, hs, W_ih, W_hh = ctx.saved_tensors
        T, B, G4 = gates.shape
        dgates = torch.empty_like(gates)
        H.lstm_bwd(dhn.float().contiguous(), gates, cs, W_hh, dgates)
        dg = dgates.view(T * B, G4)
        emb_r, wih_r, whh_r, bih_r, bhh_r = ctx.leaf_refs
        # the embedding gradient FIRST (it hangs on a product of its own), with the two bias gradients in its launch (db_hh = db_ih:
        # a tensor of its own, autograd would clone a shared one); the two weight-gradient products close the chain -- this side
        # stream is as long as the conv stack's backward beside it, every launch on it counts
        db = grad_out(bih_r); db2 = grad_out(bhh_r)
        demb = dx = None
        if OPT.lstm_tail_fused:
            if ctx.needs_input_grad[1]:
                dx = dg.mm(W_ih)
                demb = grad_out(emb_r, (ctx.vocab, xs.shape[2]))
            H.lstm_bwd_tail(idx, dx, demb, dgates, db, db2)
        dW_hh = torch.mm(dg.t(), hs[:T].reshape(T * B, -1), out=grad_out(whh_r))
        dW_ih = torch.mm(dg.t(), xs.view(T * B, -1), out=grad_out(wih_r))
        if not OPT.lstm_tail_fused:
            torch.sum(dg, 0, out=db)
            db2.copy_(db)
            if ctx.needs_input_grad[1]:
                dx = dg.mm(W_ih)
                demb = grad_out(emb_r, (ctx.vocab, xs.shape[2]))
                H.embedding_bwd(idx, dx, demb)
        return None, demb, dW_ih, dW_hh, db, db2


class NllMeanFunction(torch.autograd.Function):
    """F.nll_loss(log_probs, label) with mean reduction (train.py:41): one launch forward, one backward."""

    @staticmethod
    def forward(ctx, logp, label):
        H._dev(logp, "log-probs")
        logp = logp.float().contiguous()
        label = label.long().contiguous()
        loss = torch.empty(1, dtype=torch.float32, device=logp.device)
        H.nll_mean_fwd(logp, label, loss)
        ctx.save_for_backward(label)
        ctx.shape = logp.shape
        return loss[0]

    @staticmethod
    def backward(ctx, gloss):
        (label,) = ctx.saved_tensors
        gout = torch.empty(ctx.shape, dtype=torch.float32, device=label.device)
        H.nll_mean_bwd(label, gloss.float().reshape(1).contiguous(), gout)
        return gout, None


def nll_loss_mean(logp, label):
    """Drop-in for F.nll_loss(logp, label) (mean) on GPU tensors; falls back to torch elsewhere."""
    if logp.is_cuda and logp.dim() == 2 and OPT.fused_nll:
        return NllMeanFunction.apply(logp, label)
    return torch.nn.functional.nll_loss(logp, label)
