"""Relational-layer hot path (model.py:104-162 of the reference) as ONE
torch.autograd.Function over the C-ABI HIP kernels.

chain path ("f16s", the module default on the *-fp configs):
  forward : tables of the factored first layer -> ONE g_theta chain launch (pair sum on chip) -> f_phi + log_softmax (+ loss)
  backward: f_phi grads -> ONE backward-chain launch (ReLU gates from lane masks, pair-axis reductions on chip) -> dx, dq;
            the three 256 x 256 weight gradients as one launch on a side stream, dW_0 from the reductions
per-layer path ("bf16" / "fp32": the *-sd models, hooks, exact-fp32 parity):
  forward : K1 pair build -> one GEMM (bias + ReLU fused) per g layer -> pair sum -> f_phi
  backward: pair-sum broadcast + ReLU gate -> per layer {wgrad, dgrad + gate} -> pair-axis reductions -> dx, dq

PyTorch only allocates buffers and supplies the stream; all arithmetic runs in
librn_hip.so.  There is no CPU / eager fallback."""
from __future__ import annotations

import os
import torch

from . import rn_hip as H
from .options import OPT

PRECISIONS = ("bf16", "f16s", "fp32", "bf16x3")
# "f16s2": what the module turns "f16s" into for a forward pass in eval() mode (options.eval_two_pass) -- the chain path with hi + lo
# split weights on every layer when nothing needs a gradient (batch-position-invariant log-probs); with a gradient: plain "f16s"
_INTERNAL_PRECISIONS = PRECISIONS + ("f16s2",)


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
    """MFMA-operand copies of the g weights, re-packed only when a weight's version counter moves.
    chain=True  (the register-resident chains, "f16s"): fragment-major images -- layer 0 hi + lo fp16 images of W0[:, 0:k], layers
                1..3 F16S_DITHER tile-dithered fp16 images each (an injected layer: its H columns only), the backward chain's bf16
                W^T images, W0^T in fp32 for the table kernel -- all in ONE launch (rn_pack_matrix_frag_many);
    chain=False (the per-layer kernels, bf16 / fp32 storage): row-major (N, Kpad) forward copies and, for l >= 1, transposed
                (G_{l-1}, N) ones for dgrad.
    Both: (in, out) fp32 copies of the f_phi weights (coalesced reads in the one-launch f_phi kernels)."""

    def __init__(self):
        self.key = None
        self.h8 = True                             # e4m3 activation copies for this module (and options.h8); a trainer's copy guard clears it
        self.fwd, self.bwd = [], []
        self.fT = None                             # transposed fp32 copies of the f_phi weights
        self.w0T = None                            # W_0^T in fp32: the table kernel of the factored first layer
        self._last = None                          # arguments of the previous get(): what repack_ahead() repeats
        self._ahead = None                         # key of an ahead-of-time pack not consumed yet
        self._fphi_sync = None                     # sync workspace of the feature-split f_phi launch (rn_f_phi_split), one per module
        self.q_grad_async = False                  # the question is QuestionLSTMFunction's output on a side stream: its gradient may be handed over by event (_GRAD_EVENTS)
        self.frag_hi, self.frag_lo = [], []        # fragment-major fp16 hi / lo images of the forward chain
        self.fragT = []                            # backward step s -> W_{L-1-s}^T (bf16 fragment-major)

    @staticmethod
    def _key(code, chain, bwd_images, g_w, f_w):
        return (code, chain, bwd_images, tuple((w.data_ptr(), w._version) for w in list(g_w) + list(f_w or ())))

    def get(self, plan: LayerPlan, g_w, code, chain=False, bwd_images=True, f_w=None):
        """chain = True: the training / dithered image set; chain = "two_pass": plain hi + lo images of every layer (the
        batch-invariant inference arithmetic, rn_g_chain_fwd_rr_f16s_alg0 with dither 0; never with bwd_images)."""
        key = self._key(code, chain, bwd_images, g_w, f_w)
        self._last = (plan, tuple(g_w), code, chain, bwd_images, tuple(f_w) if f_w is not None else None)
        if self._ahead == key:                              # packed by repack_ahead() earlier in this forward pass
            self._ahead = None
            return self.fwd, self.bwd
        self._ahead = None
        # while a hipGraph is being captured the pack kernels must be part of it (a replay sees
        # new weights every step), so the cache is bypassed
        if key == self.key and not torch.cuda.is_current_stream_capturing():
            return self.fwd, self.bwd
        return self._pack(key, plan, g_w, code, chain, bwd_images, f_w)

    def fphi_sync(self, device):
        if self._fphi_sync is None or self._fphi_sync.device != torch.device(device):
            self._fphi_sync = H.f_phi_split_sync_ws(device)
        return self._fphi_sync

    def repack_ahead(self):
        """Repeat the previous get()'s pack NOW, on the caller's current stream -- RN.forward calls this on the question
        encoder's side stream, whose fork and join around the conv stack exist anyway (a fork / join of its own costs
        more than the 11 us it hides: measured).  The next get() with the same arguments returns the images without
        launching anything; the caller's join orders it after this stream."""
        if self._last is None:
            return
        plan, g_w, code, chain, bwd_images, f_w = self._last
        if not torch.is_grad_enabled():
            bwd_images = False
        key = self._key(code, chain, bwd_images, g_w, f_w)
        if key != self.key or torch.cuda.is_current_stream_capturing():
            self._pack(key, plan, g_w, code, chain, bwd_images, f_w)
        self._ahead = key

    def _pack(self, key, plan, g_w, code, chain, bwd_images, f_w):
        dt = H.torch_dtype(code)
        dev = g_w[0].device
        self.fwd, self.bwd, self.fragT, self.frag_hi, self.frag_lo = [], [], [], [], []
        self.w0T = None
        frag_jobs = []
        k, inj = plan.k, plan.inject
        for l, w in enumerate(g_w):
            N, kt = w.shape
            assert kt == plan.ktrue[l] and N == plan.widths[l], (w.shape, plan.ktrue[l], plan.widths[l])
            wc = w.detach()
            if not wc.is_contiguous():
                wc = wc.contiguous()
            if chain:
                # columns that enter the MFMA image: the factored first layer keeps W0[:, 0:k], an injected layer W[:, 0:G_prev]
                kimg = k if l == 0 else (plan.widths[l - 1] if l == inj else kt)
                two = chain == "two_pass"
                V = 1 if (l == 0 or two) else H.F16S_DITHER
                wh = torch.empty(V, 256 * 256, dtype=torch.float16, device=dev)
                frag_jobs.append((wc, kt, 1, N, kimg, wh, 4 | int(l == 0) | ((V << 8) if V > 1 else 0)))
                self.frag_hi.append(wh)
                if l == 0 or two:
                    wl = torch.empty(256 * 256, dtype=torch.float16, device=dev)
                    frag_jobs.append((wc, kt, 1, N, kimg, wl, 8 | int(l == 0)))
                    self.frag_lo.append(wl)
                self.fwd.append(None)
                self.bwd.append(None)
                continue
            wp = torch.empty(N, plan.kpad[l], dtype=dt, device=dev)
            H.pack_matrix(wc, kt, 1, N, kt, wp, code, plan.kpad[l], N)
            self.fwd.append(wp)
            if l >= 1 and bwd_images:
                gp = plan.widths[l - 1]           # only the H_{l-1} columns take part in dgrad
                wt = torch.empty(gp, N, dtype=dt, device=dev)
                H.pack_matrix(wc, 1, kt, gp, N, wt, code, N, gp)      # wt[k][n] = w[n][k]
                self.bwd.append(wt)
            else:
                self.bwd.append(None)
        if chain:
            if bwd_images:
                self.fragT = list(torch.empty(plan.L - 1, 256 * 256, dtype=torch.bfloat16, device=dev))     # equally spaced (rn_g_chain_bwd_rr)
                for st, wf in enumerate(self.fragT):
                    wc = g_w[plan.L - 1 - st].detach().contiguous()
                    frag_jobs.append((wc, 1, wc.shape[1], 256, 256, wf, st == 0))      # element (in, out) = W[out][in]
            w0 = g_w[0].detach().contiguous()
            self.w0T = torch.empty(w0.shape[1], w0.shape[0], dtype=torch.float32, device=dev)
            frag_jobs.append((w0, w0.shape[1], 1, w0.shape[0], w0.shape[1], self.w0T, 2))
        # f_phi weights as (in, out) fp32 copies: the forward kernel's thread-per-output-feature walk is then coalesced
        self.fT = None
        if f_w is not None:
            self.fT = [torch.empty(w.shape[1], w.shape[0], dtype=torch.float32, device=dev) for w in f_w]
            small = all(max(w.shape) <= 256 for w in f_w)           # (the one-launch packer takes matrices up to 256 x 256)
            for w, wt in zip(f_w, self.fT):
                wc = w.detach().contiguous()
                if small:
                    frag_jobs.append((wc, wc.shape[1], 1, wc.shape[0], wc.shape[1], wt, 2))
                else:
                    # the wide f_phi of the state-description models (512 / 1024): its forward runs one feature-split launch per
                    # layer on these copies (rn_small.hip, fp_wide_*) instead of pulling 3 MB of weights through one CU per 4 rows
                    H.pack_matrix(wc, 1, wc.shape[1], wc.shape[1], wc.shape[0], wt, H.RN_F32, wc.shape[0], wc.shape[1])
        if frag_jobs:
            H.pack_matrix_frag_many(frag_jobs)                                     # one launch for all images
        self.key = key
        return self.fwd, self.bwd


# Gradients handed to ANOTHER autograd node on ANOTHER stream without making the producing node's stream wait for them: data_ptr ->
# event.  The one user: the question gradient of a question-injected relational layer (ir-*), which comes out of the weight-gradient
# launch on its side stream ~100 us after dx is ready -- QuestionLSTMFunction.backward (the question encoder's stream) waits for the
# event itself, and the conv stack's backward starts with dx instead of behind dq (RN.forward sets PackedWeights.q_grad_async when
# the question IS that node's output; any other producer gets the synchronous hand-off).
_GRAD_EVENTS = {}
HANDED_BY_EVENT = [0]                              # how many gradients went that way (tests)
_SIDE_STREAMS = {}
# Launch-order choices of the backward pass that only measurements decide (the captured graph's queue order): kept as named knobs so that
# tools/dbg/exp_bench.py can A/B them on one box.  wgrad_late: 0 the g_theta weight-gradient launch right behind the backward chain,
# 1 behind the partial sums AND the layer-0 stream, 2 behind dx / dq too, 3 behind the partial sums only; -1 (default, round 6) =
# 0 for models without question injection, 1 for the injected ones (ir-*: their question gradient comes off this stream) -- with the
# wide-unit kernel on 160 workgroups the early launch is over before the conv stack's heavy last kernels start: original-fp
# 97.8 -> 100.4 k q/s, the 14 x 14 stress shape 13.4 -> 14.05 k, B = 640 +-0; ir-fp 91.2 -> 82.7 k with it, hence not there
# (profiles/r06_ablations/ab_kb_total_wide2.txt);
# conv_wgrad_stream: the side stream the conv weight gradients share (0 the g_theta weight gradient's, 2 the layer-0 stream's);
# fphi_grads_late: f_phi's parameter gradients on the layer-0 stream instead of in front of the backward chain;
# chain_balance: the reducing backward chain's units beyond a whole number of rounds over the CUs run tile by tile (0: whole units only);
# dq_async: the question gradient of a question-injected layer is handed to the question encoder's backward by event (0: the main stream waits for it);
# lstm_hn_view / text_first (round 6, both measured and left off: profiles/r06_ablations/ab_lstm_hn_view.txt, ab_text_overlap.txt): the
# question encoder's output as a view instead of a clone (one memcpy node less in front of the join with the conv stack: 103.9 vs
# 104.0 k q/s), the question encoder captured in front of the conv stack instead of behind it (98.6 vs 100.2 k; ir-fp 87.5 vs 90.9 k).
SCHED = {"lstm_hn_view": 0, "text_first": 0, "wgrad_late": -1, "conv_wgrad_stream": 2, "fphi_grads_late": 1, "chain_balance": 1, "dq_async": 1}


def _dev_key(dev):
    dev = torch.device(dev)
    return dev.index if dev.index is not None else torch.cuda.current_device()


def fresh_stream(dev):
    """A pool stream that is none of this package's role streams.  torch.cuda.Stream() hands out its 32 pool streams round-robin:
    a process that builds many models / trainers gets the SAME hipStream again 32 allocations later, and two roles of one
    captured step on one hipStream -- the question encoder's stream and the weight-gradient stream it waits on by event -- make
    the capture's fork / join topology cyclic: hip::Stream::EndCapture then recurses until the stack ends (round 6: a segmentation
    fault at the 16th ir-fp trainer of one process, found by the multi-seed convergence tool)."""
    taken = {s.cuda_stream for (d, _w), s in _SIDE_STREAMS.items() if d == _dev_key(dev)}
    taken.add(torch.cuda.current_stream(dev).cuda_stream)
    cap = getattr(torch.cuda.graph, "default_capture_stream", None)        # (torch's class-level capture stream, once it exists)
    if cap is not None:
        taken.add(cap.cuda_stream)
    for _ in range(64):
        s = torch.cuda.Stream(device=dev)
        if s.cuda_stream not in taken:
            return s
    raise RuntimeError("no HIP stream left that is not one of the package's role streams")


def _side_stream(dev, which=0):
    """THE stream of role `which` on `dev`, one per process (0, 1, 2: the backward pass' side streams; "text": the question encoder's;
    "warmup": the trainer's eager passes in front of a capture): roles never share a hipStream (fresh_stream), instances do."""
    key = (_dev_key(dev), which)
    s = _SIDE_STREAMS.get(key)
    if s is None:
        s = _SIDE_STREAMS[key] = fresh_stream(dev)
    return s


# ---- gradients straight into a FlatGradBucket (dp.py) ---------------------------------------------------------------
# A bucket registers (parameter -> slot of its flat buffer).  A backward function that is about to allocate the gradient of a
# registered parameter takes a FRESH VIEW of the slot instead and lets its kernels write there: autograd then assigns that
# view as .grad (no copy: a fresh view is uniquely owned), and the bucket's gather step finds the gradient already in place --
# no concatenation kernel between the end of the backward pass and the optimiser.  Only when autograd will ASSIGN
# (_assign_only: .grad is None, no hooks): accumulating `grad += view-of-the-same-memory` would double the new gradient.
_GRAD_SLOTS = {}
_SLOTS_OUT = set()       # parameters whose slot has been handed out in the backward pass that is running (cleared by an engine callback)


def register_grad_slots(params, flat, offsets):
    import weakref
    for p_, o in zip(params, offsets):
        _GRAD_SLOTS[id(p_)] = (weakref.ref(p_), weakref.ref(flat), o)


def reset_grad_slot_handouts():
    """Forget the slot hand-outs of a backward pass that never finished (it raised part-way: the engine callback that clears them
    did not run, and with a non-empty set no later pass would queue one).  The trainer calls this before every backward pass."""
    _SLOTS_OUT.clear()


def grad_out(param, shape=None):
    """The tensor a backward function writes `param`'s gradient into (see above); shape: the gradient's shape when the caller
    needs another one than the parameter's (then no slot is used unless the element count matches a contiguous view)."""
    shape = tuple(param.shape) if shape is None else tuple(shape)
    slot = _GRAD_SLOTS.get(id(param)) if OPT.grads_in_bucket else None
    if slot is not None:
        pref, fref, off = slot
        flat = fref()
        if (pref() is param and flat is not None and flat.device == param.device and shape == tuple(param.shape) and _assign_only(param)
                and id(param) not in _SLOTS_OUT):
            # ONE hand-out per parameter and backward pass: a parameter that receives two gradients in a pass (the model run twice,
            # the losses summed) would get the same memory twice, the second function overwriting what autograd still holds as the
            # first addend (-> 2 x the second gradient, silently).  The second request gets a tensor of its own; autograd adds.
            if not _SLOTS_OUT:
                try:
                    torch.autograd.Variable._execution_engine.queue_callback(_SLOTS_OUT.clear)
                except RuntimeError:                          # (not inside a backward pass: nothing to protect)
                    return flat[off:off + param.numel()].view(shape)
            _SLOTS_OUT.add(id(param))
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

    def __init__(self, masks, gate=False):
        self.masks = masks
        self.gate = gate          # the forward chain wrote the last layer's gate into the sign bits of the e4m3 H_{L-2} image


def padded_j(n):
    """Pair rows per (question, i) group on the chain path: n itself when a wave's 32 rows fit (n % 32 == 0), else the next
    multiple of 32 -- the chain then runs on a PADDED j axis (include/rn_hip.h, rn_g_chain_fwd_rr_f16s_alg0; the 14 x 14 grid:
    196 -> 224) -- or None when the kernel does not cover n either (n % 4 != 0)."""
    if n % 32 == 0:
        return n
    return _ru(n, 32) if n % 4 == 0 else None


def chain_ok(plan: LayerPlan, B, n):
    """ONE shape predicate for the register-resident chains (rn_chain_rr.hip; the "f16s" arithmetic, what `precision: auto`
    resolves to) -- shared by the module, the autograd function and bench.py: exactly four 256-wide g layers, at most 32 features
    per object, and the question injected
      at layer 0 (config.json original-fp): any n % 4 == 0 (n % 32 != 0 runs on the padded j axis), whole 256-row tiles;
      at layer 2 (ir-fp): whole waves per (question, i) and whole tiles per question (n % 32 == 0, n*n % 256 == 0).
    Everything else -- the 512-wide *-sd models, other depths, forward hooks -- runs the per-layer kernels (bf16 / fp32)."""
    if not (plan.L == 4 and all(w == 256 for w in plan.widths) and plan.k <= 32):
        return False
    T = H.g_chain_rr_tile()
    if plan.inject == 0:
        njp = padded_j(n)
        return njp is not None and (B * n * njp) % T == 0
    return plan.inject == 2 and plan.ktrue[2] == 256 + plan.Q and n % 32 == 0 and (n * n) % T == 0


f16s_ok = chain_ok


def _h_copy_dtype(packed=None):
    """Storage type of the H_0..2 copies the forward chain keeps for the weight gradient (row-blocked images, rn_g_wgrad_blocked
    is their only reader): OCP e4m3 bytes -- half the bytes written and read back -- or bf16 with options.h8 off (A/B
    measurements, error comparisons) or with the MODULE's own switch off (`packed.h8 = False`: what a trainer's copy guard
    falls back to -- scoped to the module it watched, not to the process)."""
    return torch.float8_e4m3fn if (OPT.h8 and getattr(packed, "h8", True)) else torch.bfloat16


def alg0_wgrad_ok(plan, k):
    """Layer-0 weight gradient from the pair reductions (rn_wgrad0_from_reductions) instead of a pass over dZ_0 and P."""
    return plan.inject == 0 and k <= 32


def _tables(x, q, plan, g_b, w0T, inj_w, B, n, k, Q, G, coord=None):
    """Tables of the factored first layer (+ the question rows of an injected later layer): -> (Xp, Vc, Vq | None).
    coord (2, n): x is the conv grid itself (kf = k - 2 columns), the coordinate tags are read from the table in the kernel."""
    dev = x.device
    Xp = torch.empty(B * n + 1, 64, dtype=torch.float16, device=dev)   # (+ the all-zero object row the padded-j chain reads for j >= n)
    if n % 32:                                                   # (only the padded-j chain reads the row: no fill launch otherwise)
        Xp[B * n].zero_()
    Vc = torch.empty(B * n, G, dtype=torch.float32, device=dev)
    if inj_w is None:
        H.pair_tables(x, q, w0T, g_b[0], Xp, Vc, B, n, k, Q, G, coord=coord)
        return Xp, Vc, None
    inj = plan.inject
    H.pair_tables(x, None, w0T, g_b[0], Xp, Vc, B, n, k, 0, G, coord=coord)
    Gp = plan.widths[inj - 1]
    Vq = torch.empty(B, G, dtype=torch.float32, device=dev)
    # Vq[b, f] = b_inj[f] + sum_c q[b, c] W_inj[f, Gp + c]   (model.py:135-141: the question is the layer's trailing Q columns)
    H.gemm_f32(q, Q, 1, inj_w, 1, inj_w.shape[1], Vq, G, B, G, Q, bias=g_b[inj], b_off=Gp)
    return Xp, Vc, Vq


def chain_forward(x, q, plan: LayerPlan, g_b, packed, keep, inj_w=None, coord=None, two_pass=False):
    """The chain path's forward (model.py:108-152): tables of the factored first layer + ONE launch for the four g layers and the
    pair sum.  -> (Hs, masks, xg, njp): Hs = the row-blocked copies of H_0..2 (None: inference), masks = RRMasks, xg = the pair
    sums (PairSumPartials: the f_phi launch adds the per-tile partials up), njp = pair rows per (question, i) group."""
    B, n, k = x.shape
    if coord is not None:
        k += coord.shape[0]
    Q, G, L, R = q.shape[1], plan.widths[-1], plan.L, H.g_chain_rr_tile()
    dev = x.device
    Xp, Vc, Vq = _tables(x, q, plan, g_b, packed.w0T, inj_w, B, n, k, Q, G, coord)
    njp = n if inj_w is not None else padded_j(n)          # pair rows per (question, i) group: padded where n % 32 != 0
    Mp = B * n * njp
    masks = Hs = None
    gate = False
    if keep:
        Hs = [torch.empty(Mp, G, dtype=_h_copy_dtype(packed), device=dev) for l in range(L - 1)] + [None]
        masks = list(torch.empty(L, H.g_chain_rr_mask_bytes(Mp), dtype=torch.uint8, device=dev))
        # the last layer's weight-gradient gate job reads the gate from the sign bits of the e4m3 H_{L-2} image
        gate = Hs[0].dtype in H.FP8_DTYPES and (n * njp) % 64 == 0
    lo = packed.frag_lo if two_pass else packed.frag_lo[0]          # (two_pass: hi + lo on every layer; inference only)
    assert not (two_pass and keep)
    if njp != n:
        part = torch.empty(Mp // R * 2, G, dtype=torch.float32, device=dev)      # two partial rows per tile (it may straddle questions)
        H.g_chain_fwd_rr_f16s_alg0(Xp, Vc, n, packed.frag_hi, lo, g_b, Hs, masks, part, Mp, G, njp=njp, gate=gate)
        xg = torch.empty(B, G, dtype=torch.float32, device=dev)
        H.pair_sum_tiles(part, xg, Mp, n * njp, G)
    else:
        part = torch.empty(Mp // R, G, dtype=torch.float32, device=dev)
        H.g_chain_fwd_rr_f16s_alg0(Xp, Vc, n, packed.frag_hi, lo, g_b, Hs, masks, part, Mp, G, Vq=Vq,
                                   inject=plan.inject if inj_w is not None else 0, gate=gate)
        xg = PairSumPartials(part, B, (n * n) // R, G)
    return (Hs[:-1] if Hs is not None else None), (RRMasks(masks, gate) if masks is not None else None), xg, njp


def layers_forward(x, q, plan: LayerPlan, g_b, wfwd, code, keep_inputs=True, layer_hook=None, stop_at=None, gcode=None):
    """The per-layer path (bf16 / fp32 storage; model.py:108-145): K1 pair build, then one fused GEMM + bias + ReLU launch per g
    layer (the question broadcast into the trailing columns of an injected layer's input).  Returns (inputs, H_L): the list of layer
    INPUT buffers [A_0 .. A_{L-1}] and the last activation.  layer_hook(l, A_l, H_out) is called after every layer (the
    hook-compatible path of extract.py:43,101); stop_at = l: return after the INPUT of layer l exists."""
    B, n, k = x.shape
    Q = q.shape[1]
    M = B * n * n
    dt = H.torch_dtype(code)
    gcode = code if gcode is None else gcode            # the GEMMs' arithmetic ("bf16x3": RN_F32X3 on fp32 storage)
    dev = x.device
    inj = plan.inject
    ld0 = plan.kpad[0]
    P = torch.empty(M, ld0, dtype=dt, device=dev)
    H.pair_build_fwd(x, q if inj == 0 else None, P, code, B, n, k, Q if inj == 0 else 0, ld0)
    if stop_at == 0:
        return [P], None
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
        H.g_linear_fwd(cur, plan.kpad[l], wfwd[l], plan.kpad[l], g_b[l], out, ldh, gcode, M, N, plan.kpad[l])
        if nxt_wide:
            H.qst_broadcast(q, out, code, B, n, Q, N, ldh)
        if layer_hook is not None:
            layer_hook(l, cur, out)
        if stop_at is not None and l + 1 == stop_at:         # the caller wants the INPUT of layer stop_at only
            return inputs + [out], None
        if l + 1 < plan.L:
            inputs.append(out)
        if not keep_inputs and l >= 1:
            inputs[l] = None
        cur = out
    return inputs, cur


class PairSumPartials:
    """The forward chain's per-tile partial pair sums, not yet added up: (B * parts, G) fp32.  The f_phi launch adds them itself
    (rn_f_phi_fwd_from_partials) and writes the (B, G) sums into `.xg`."""

    def __init__(self, part, B, parts, G):
        self.part, self.B, self.parts, self.G = part, B, parts, G
        self.xg = torch.empty(B, G, dtype=torch.float32, device=part.device)


def f_phi_forward(xg, fw, fb, mask, wT=None, label=None, pre_bwd=False, packed=None):
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
    if OPT.fphi_split and packed is not None and H.f_phi_split_ok(B, G, F1, F2, A) and (wT is not None or not pre_bwd):
        # the feature-split MFMA chain (rn_fphi.hip): one launch, 16 workgroups, in-launch hand-offs -- every shape of config.json's
        # 256-wide f_phi at B <= 64 (the row-split kernels below: everything else)
        sync = packed.fphi_sync(dev)
        loss = torch.empty((), dtype=torch.float32, device=dev) if label is not None else None
        part, parts = (lazy.part, lazy.parts) if lazy is not None else (None, 0)
        if pre_bwd and label is not None:
            dxg = torch.empty(B, G, dtype=torch.float32, device=dev)
            ws = H.f_phi_split(part, parts, xg, fw, fb, wT, mask, label, f1, f2, out, loss, sync, dxg=dxg)
            return f1, f2, out, loss, (ws, dxg)
        H.f_phi_split(part, parts, xg, fw, fb, None, mask, label, f1, f2, out, loss, sync)
        return (f1, f2, out, loss) if label is not None else (f1, f2, out)
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
    the kernels and the input gradient comes back in the grid's own layout (no concatenation, no slicing).

    Two paths, chosen by `precision` alone:
      "f16s"          the CHAIN path (chain_ok shapes): tables of the factored first layer, one forward-chain launch, one
                      backward-chain launch (pair-axis reductions on chip), one launch for the three 256 x 256 weight gradients;
      "bf16" / "fp32" the PER-LAYER path: pair matrix (K1), one GEMM launch per layer and direction."""

    @staticmethod
    def forward(ctx, x, q, mask, plan, packed, precision, label, coord, *params):
        # (precision, grad mode of the CALLER): inside forward() autograd is always off and ctx.needs_input_grad ignores torch.no_grad()
        precision, grad_on = precision if isinstance(precision, tuple) else (precision, True)
        ctx.set_materialize_grads(False)
        q_grad_async, packed.q_grad_async = bool(getattr(packed, "q_grad_async", False)), False   # ONE call's permission (RN.forward sets it right in front of this call)
        L = plan.L
        g_w, g_b = params[0:L], params[L:2 * L]
        f_w, f_b = params[2 * L:2 * L + 3], params[2 * L + 3:2 * L + 6]
        H._dev(x, "x")
        H._dev(q, "qst")
        chain = precision in ("f16s", "f16s2")
        code = H.RN_BF16 if chain else H.dtype_code(precision)
        gcode = H.RN_F32X3 if precision == "bf16x3" else code     # arithmetic of the per-layer GEMMs (storage: code)
        x = x.float() if x.dtype != torch.float32 else x
        q = q.float().contiguous() if (q.dtype != torch.float32 or not q.is_contiguous()) else q
        B, n, k = x.shape
        if coord is not None:
            k += coord.shape[0]
        Q = q.shape[1]
        M = B * n * n
        dev = x.device
        need_grad = grad_on and any(ctx.needs_input_grad)      # under no_grad nothing is kept for a backward pass (inference kernels)
        if chain and not chain_ok(plan, B, n):
            raise RuntimeError('precision "f16s" needs the register-resident chains: four 256-wide g layers, <= 32 features per object, '
                               'the question at layer 0 (n % 4 == 0) or 2 (n % 32 == 0), whole 256-row tiles (functional.chain_ok); '
                               'use "auto", "bf16" or "fp32" here')
        if coord is not None and not chain:
            raise RuntimeError("internal: a coordinate table was passed but the chain path does not apply (grid_path_ok)")
        two_pass = precision == "f16s2" and not need_grad
        wfwd, wbwd = packed.get(plan, g_w, code, chain=("two_pass" if two_pass else chain), bwd_images=need_grad, f_w=f_w)
        gb = [b.detach().contiguous() for b in g_b]
        G = plan.widths[-1]
        njp = n
        if chain:
            inj_w = None
            if plan.inject > 0:
                inj_w = g_w[plan.inject].detach()
                inj_w = inj_w if inj_w.is_contiguous() else inj_w.contiguous()
            Hs, HL, xg, njp = chain_forward(x, q, plan, gb, packed, need_grad, inj_w=inj_w, coord=coord, two_pass=two_pass)
            inputs = [None] + Hs if Hs is not None else [None] * L
            if COPY_HEALTH_PROBE is not None and need_grad:
                for l in range(1, L):                              # inputs[l] = the copy of H_{l-1}
                    if inputs[l].dtype in H.FP8_DTYPES:
                        COPY_HEALTH_PROBE.append((l - 1, H.fp8_copy_health(HL.masks[l - 1], inputs[l], inputs[l].numel() // G)))
        else:
            inputs, HL = layers_forward(x, q, plan, gb, wfwd, code, keep_inputs=need_grad, gcode=gcode)
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
            r = f_phi_forward(xg, fw, fb, mask, wT=packed.fT, label=label, packed=packed,
                              pre_bwd=need_grad and OPT.fphi_fused_bwd and (isinstance(xg, PairSumPartials) or OPT.fphi_split))
            f1, f2, out, loss = r[:4]
            if len(r) == 5:
                ctx.fphi_pre = r[4]
        else:
            f1, f2, out = f_phi_forward(xg, fw, fb, mask, wT=packed.fT, packed=packed)
        if isinstance(xg, PairSumPartials):
            xg = xg.xg
        ctx.label = label
        if need_grad:
            ctx.plan, ctx.code, ctx.dims = plan, code, (B, n, k, Q, M, G, F1, F2, A)
            ctx.gcode = gcode
            ctx.chain, ctx.njp = chain, njp
            ctx.q_grad_async = q_grad_async
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
        ctx.fphi_job = None
        if gout is None and ctx.fphi_pre is not None and is_unit_loss_grad(gloss):
            # the forward launch already ran the dz chain for d loss = 1: the parameter gradients are all that is left
            ws_, dxg = ctx.fphi_pre
            job = lambda: H.f_phi_bwd_grads(ws_, xg, f1, f2, (dW1, dW2, dW3), (db1, db2, db3))
            if ctx.chain and SCHED["fphi_grads_late"]:
                # Nothing in the backward pass reads these six gradients, and no launch of the replayed step lasts less than 6-7 us:
                # in front of the backward chain this one sat on the critical path.  The chain path runs it on the layer-0 stream
                # in front of that weight gradient -- a fork that exists anyway (a fork / join pair of its own cost more than the
                # kernel): +0.6 % q/s (93.9 vs 93.3 k, six alternating runs)
                ctx.fphi_job = (job, [ws_, xg, f1, f2])
            else:
                job()
        elif gout is None:
            dxg = torch.empty(B, G, **f32)
            H.f_phi_bwd_nll(gloss, label, out, f2, f1, xg, fw, ctx.mask, (dW1, dW2, dW3), (db1, db2, db3), dxg)
        else:
            dxg = torch.empty(B, G, **f32)
            H.f_phi_bwd(gout, out, f2, f1, xg, fw, ctx.mask, (dW1, dW2, dW3), (db1, db2, db3), dxg)
        # ---- g_theta backward
        if ctx.chain:
            dx, dq, gW, gB = RelationalFunction._backward_chain(ctx, dxg, x, q)
        else:
            dx, dq, gW, gB = RelationalFunction._backward_layers(ctx, dxg, x, q)
        ctx.inputs = ctx.HL = None
        grads = [dx if ctx.needs_input_grad[0] else None, dq if ctx.needs_input_grad[1] else None, None, None, None, None, None, None]
        grads += gW + gB + [dW1, dW2, dW3, db1, db2, db3]
        return tuple(grads)

    # ------------------------------------------------------------------------------------------------ chain path
    @staticmethod
    def _backward_chain(ctx, dxg, x, q):
        """Backward of the chain path (autograd of model.py:108-152): ONE backward-chain launch -- with the pair-axis reductions of
        layer 0's gradient formed on chip where n % 32 == 0 --, the three 256 x 256 weight gradients as ONE launch on a side
        stream, the layer-0 weight gradient from the reductions on a stream of its own, dx / dq in one launch."""
        plan = ctx.plan
        B, n, k, Q, M, G = ctx.dims[:6]
        L, dev = plan.L, x.device
        f32 = dict(dtype=torch.float32, device=dev)
        inputs, masks, gated, g_w = ctx.inputs, ctx.HL.masks, bool(ctx.HL.gate), ctx.g_w
        njp = ctx.njp                              # pair rows per (question, i) group: > n on the padded j axis
        Mc = B * n * njp                           # ... and the pair rows the chains / weight gradients work on
        inj = plan.inject > 0                      # the question entered at layer plan.inject as a per-question bias row
        dt = torch.bfloat16
        # the last layer's gradient dZ_{L-1} = gate x dxg[question] is never stored when the copies are e4m3: its only reader
        # besides the chain itself, the layer's wgrad, takes the gate from the sign bits of its e4m3 input image (rn_g_wgrad_blocked);
        # layer 0's gradient is read by the pair-axis reductions only: formed inside the chain, dZ_0 never exists either
        red_parts = None
        if gated:
            tpu = H.g_chain_bwd_rr_red_tpu(Mc, n, njp) if OPT.chain_reduce else 0
            if tpu > 0:
                dZs = [None] + list(torch.empty(L - 2, Mc, G, dtype=dt, device=dev)) + [None]
                # balanced tail (round 5): the units beyond a whole number of rounds over the CUs run tile by tile
                whole = H.g_chain_bwd_rr_red_whole(Mc, n, njp, tpu) if SCHED["chain_balance"] else H.g_chain_bwd_rr_red_units(Mc, n, njp, tpu)
                red_parts = (torch.empty(H.g_chain_bwd_rr_red_records(Mc, n, njp, tpu, whole), 32, G, **f32), torch.empty(Mc // 16, G, **f32),
                             ((n + 7) // 8) // tpu, tpu, whole)
                H.g_chain_bwd_rr_red(dxg, masks, ctx.fragT, dZs, Mc, n, G, red_parts[0], red_parts[1], tpu, njp=njp, whole=whole)
            else:
                dZs = [None] + list(torch.empty(L - 1, Mc, G, dtype=dt, device=dev))
        else:
            dZs = list(torch.empty(L, Mc, G, dtype=dt, device=dev))            # dZs[s] belongs to layer L-1-s
        if red_parts is None:
            H.g_chain_bwd_rr(dxg, masks, ctx.fragT, dZs, Mc, n * njp, G)
        dZ_of = {L - 1 - s: dZs[s] for s in range(L)}
        gW, gB = [None] * L, [None] * L
        # question injected at layer > 0: its per-question sums Rq come from the wgrad kernel's per-split column sums when no split
        # straddles two questions, else from a pass over the layer's stored gradient (rn_blocked_question_sums)
        rq_splits, inj_out = 0, {}
        # The weight gradients are not needed by anything upstream: they go to a side stream and overlap the rest of this backward
        # AND the conv / LSTM backward that autograd runs next (small kernels that leave the chip mostly empty).  The main stream
        # re-joins at the end of the backward pass (engine callback).  Only when every parameter's .grad is None (assign, not
        # accumulate: autograd then launches no kernel on these tensors before the join).
        overlap = OPT.wgrad_overlap and all(_assign_only(p) for p in ctx.param_refs)
        # ir-*: the question gradient handed to its consumer BY EVENT (round 5): the injected layer's stored gradient is summed per
        # question by a pass of its own on the weight-gradient stream, in front of that launch, and the question encoder's backward
        # waits for the event -- not this node's stream, which returns dx ~100 us earlier than the per-question sums that fall out
        # of the 190-us weight-gradient launch would let it
        dq_by_event = bool(inj and overlap and ctx.q_grad_async and SCHED["dq_async"])
        if inj and not dq_by_event:
            z_ = H.wgrad_blocked_splits(M, n * n, L - 1, aligned=True)
            if z_ > 0 and z_ % B == 0 and (M // 64) % z_ == 0 and (n * n) % (M // z_) == 0:
                rq_splits = z_

        def _wgrads_blocked():
            """Layers 1..L-1 on the row-blocked images: ONE launch + one reduction launch (rn_g_wgrad_blocked).  The injected layer
            (input [H_{l-1} | q]): dW = [dZ^T H_{l-1} | Rq^T q] -- the H part is an ordinary job; with question-aligned splits Rq is
            the kernel's bias-gradient partials, dq and the question columns of dW follow at once, on this stream."""
            order = list(range(1, L))
            if inj:                                                # (its dq is waited for by the main stream: first)
                order.remove(plan.inject)
                order.insert(0, plan.inject)
            jobs, tmp = [], None
            for l in order:
                N_, kt_ = plan.widths[l], plan.ktrue[l]
                gW[l] = grad_out(ctx.param_refs[l], (N_, kt_))
                gB[l] = grad_out(ctx.param_refs[L + l], (N_,))
                dz_l = dZ_of[l]                                             # (None: the last layer without a stored gradient = a gate job)
                if inj and l == plan.inject:
                    tmp = torch.empty(N_, plan.widths[l - 1], **f32)
                    jobs.append((dz_l, inputs[l], tmp, gB[l]))
                else:
                    jobs.append((dz_l, inputs[l], gW[l], gB[l]))
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

        keep = [list(dZs), list(inputs), masks, dxg]         # operands read on a side stream stay alive until the join
        # WHEN the launch starts (round 4): its 470-MB stream is what it costs the step (the step with its requests ablated: 0.62 ms
        # instead of 0.70; with its arithmetic ablated: unchanged), and the first kernels behind the backward chain -- the partial
        # sums, dx / dq, the layer-0 weight gradient -- are one or two memory round trips each: 7 + 18 | 21 us alone, 18 + 71 | 66 us
        # beside the stream.  So the launch waits for the partial sums and the layer-0 weight gradient (its stream's join, below)
        # and dx reaches the conv stack's backward ~60 us earlier: +1..1.6 % q/s together with the conv weight gradients moved to the
        # layer-0 stream (they no longer queue behind this launch).  Behind dx / dq as well: -11 %; behind the partial sums ONLY
        # (the layer-0 stream joined after the launch): -12..-16 % -- the captured graph's queue order, not arithmetic
        # (tools/dbg/exp_bench.py wgrad_late=2 | 3).  The question-injected models took dq from the launch's per-question sums through round 4 (early
        # launch); since round 5 the sums are a pass of their own in FRONT of the late launch and dq is handed to its consumer by event
        # (dq_by_event above: +5 % on ir-fp -- with the main stream waiting for dq the same change measured -0.5 %).
        late_mode = SCHED["wgrad_late"] if SCHED["wgrad_late"] >= 0 else (1 if inj else 0)
        late = overlap and late_mode and (not inj or dq_by_event)
        if overlap:
            main, side = torch.cuda.current_stream(), _side_stream(dev)
            if not late:
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    _wgrads_blocked()
                    if rq_splits:
                        inj_out["event"] = side.record_event()

            def _join():
                cur = torch.cuda.current_stream()
                cur.wait_stream(side)
                keep.clear()
                #  a question gradient handed over by event that its consumer never took -- the gradient reached the
                # question encoder's backward as ANOTHER tensor (a second user of the question: the engine accumulated; a tensor
                # hook; retain_grad) and was read there without the wait: say so, and make this stream wait at least
                if _GRAD_EVENTS:
                    import warnings
                    for ev, _fill in _GRAD_EVENTS.values():
                        cur.wait_event(ev)
                    _GRAD_EVENTS.clear()
                    warnings.warn("the question gradient of the injected layer was handed over by event, but its consumer did not take it "
                                  "(the question has another user, a hook or retain_grad): it may have been read before it was complete -- "
                                  "set functional.SCHED['dq_async'] = 0 for such models")
            torch.autograd.Variable._execution_engine.queue_callback(_join)
        else:
            _wgrads_blocked()
        # ---- the injected layer's question gradient (ir-*), when the wgrad launch did not produce it
        dq = None
        if inj and not rq_splits:
            l = plan.inject
            N, kt = plan.widths[l], plan.ktrue[l]
            wl = g_w[l] if g_w[l].is_contiguous() else g_w[l].contiguous()
            Rq2 = torch.empty(B, N, **f32)
            dq = torch.empty(B, Q, **f32)

            def _wgrad_question(l=l, N=N, kt=kt):                   # gW[l][:, G_prev:] = Rq^T q  (bound NOW: N, kt are layer 0's further down)
                H.gemm_f32(Rq2, 1, N, q, Q, 1, gW[l], kt, N, Q, B, c_off=kt - Q)
            if dq_by_event:
                # on the weight-gradient stream, IN FRONT of that (late) launch: the sums + the small product run beside the partial
                # sums / dx on the main stream; the question encoder's backward waits for the event, this stream does not.  (Run by
                # the consumer on ITS stream instead -- a fifth concurrent branch of the replayed graph -- the executor serialised
                # the conv weight gradients behind the weight-gradient launch: no gain.)
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    H.blocked_question_sums(dZ_of[l], Rq2, M, n * n)
                    H.gemm_f32(Rq2, N, 1, wl, kt, 1, dq, Q, B, Q, N, b_off=kt - Q)
                    _GRAD_EVENTS.clear()
                    _GRAD_EVENTS[dq.data_ptr()] = (side.record_event(), None)
                    HANDED_BY_EVENT[0] += 1
                if late:
                    inj_out["question_cols"] = _wgrad_question     # (behind the weight-gradient launch, which creates gW[l])
                else:
                    with torch.cuda.stream(side):
                        _wgrad_question()
                keep.append([Rq2, q, wl, dq])
            else:
                H.blocked_question_sums(dZ_of[l], Rq2, M, n * n)       # (this layer's dZ is a row-blocked image)
                H.gemm_f32(Rq2, N, 1, wl, kt, 1, dq, Q, B, Q, N, b_off=kt - Q)   # Rq @ W[:, -Q:]
            if dq_by_event:
                pass
            elif overlap:
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    _wgrad_question()
                keep.append([Rq2, q])
            else:
                _wgrad_question()
        # ---- layer 0: its gradient reaches x, q and W_0 through the pair-axis reductions only (algebraic dP)
        N, kt = plan.widths[0], plan.ktrue[0]
        wl = g_w[0] if g_w[0].is_contiguous() else g_w[0].contiguous()
        Rj = torch.empty(B * n, N, **f32); Ri = torch.empty(B * n, N, **f32); Rq = torch.empty(B, N, **f32)
        if red_parts is not None:                                  # the chain has already reduced: add its partials up
            H.pair_reduce_parts(red_parts[0], red_parts[1], Rj, Ri, Rq, B, n, N, red_parts[2], njp=njp, tpu=red_parts[3], whole=red_parts[4])
        else:
            H.pair_reduce_bwd(dZ_of[0], N, Rj, Ri, Rq, H.RN_BF16, B, n, N, njp=njp)
        if late:
            late_ev = torch.cuda.current_stream().record_event()

        def _wgrad0():
            # dW_0 = [Rj^T X | Ri^T X | Rq^T Q], db_0 = sum_b Rq: three tiny products on (B*n)-row matrices instead of a 235 MB pass
            # over dZ_0 and P (and with fp32 x instead of P's rounded copy)
            gW[0] = grad_out(ctx.param_refs[0], (N, kt))
            gB[0] = grad_out(ctx.param_refs[L], (N,))
            H.wgrad0_from_reductions(Rj, Ri, Rq, x, q if plan.inject == 0 else None, gW[0], gB[0], coord=ctx.coord)
        if overlap:
            # a stream of its own (measured on one box: on the weight-gradient stream 85.6, on the main stream behind dx / dq
            # 87.0, here 87.9 k q/s; forked behind the dx / dq launch instead of in front of it: -7 %); the weight-gradient
            # stream waits for it, the end-of-backward join for that one
            s0 = _side_stream(dev, 2)
            s0.wait_stream(main)
            with torch.cuda.stream(s0):
                if ctx.fphi_job is not None:
                    ctx.fphi_job[0]()
                    keep.append(ctx.fphi_job[1])
                    ctx.fphi_job = None
                _wgrad0()
            if not (late and late_mode == 3):
                side.wait_stream(s0)
            # x and q too: they are alive only through this node's saved tensors, which autograd releases as soon as
            # backward() returns -- the caching allocator would hand their blocks to the conv / LSTM backward that the
            # main stream runs next while this side-stream kernel still reads them (seen as a wrong dW_0 on a busy GPU)
            keep.append([Rj, Ri, Rq, x, q, ctx.coord, red_parts])
        else:
            if ctx.fphi_job is not None:
                ctx.fphi_job[0]()
                ctx.fphi_job = None
            _wgrad0()
        if ctx.coord is not None:
            # the gradient goes straight into the conv grid's layout (B, k - 2, n): the two coordinate columns carry
            # none (model.py:216) and autograd's view-backward hands the conv stack a contiguous tensor
            dx = torch.empty(B, x.shape[2], n, **f32).permute(0, 2, 1)
        else:
            dx = torch.empty(B, n, k, **f32)
        if plan.inject == 0:
            dq = torch.empty(B, Q, **f32)
            H.pair_dx_dq(Rj, Ri, Rq, wl, dx, dq, B, n, k, Q, N)                    # dx and dq in one launch
        else:
            H.pair_dx_dq(Rj, Ri, None, wl, dx, None, B, n, k, 0, N)                # (dq came from the injected layer)
        if late:
            if late_mode == 2:
                side.wait_stream(main)                             # behind dx / dq
            else:
                side.wait_event(late_ev)                           # behind the partial sums only
            with torch.cuda.stream(side):
                _wgrads_blocked()
                if "question_cols" in inj_out:
                    inj_out["question_cols"]()
            if late_mode == 3:
                side.wait_stream(s0)
        if rq_splits:
            dq = inj_out["dq"]
            if "event" in inj_out:                                 # produced on the wgrad stream
                if ctx.q_grad_async and SCHED["dq_async"]:
                    # its consumer waits for it (QuestionLSTMFunction.backward); this stream goes on with dx: the conv stack's
                    # backward starts ~100 us earlier (ir-fp: +x % on the step)
                    _GRAD_EVENTS.clear()
                    _GRAD_EVENTS[dq.data_ptr()] = (inj_out["event"], None)
                else:
                    torch.cuda.current_stream().wait_event(inj_out["event"])
                    dq.record_stream(torch.cuda.current_stream())
                keep.append(inj_out["keep"])
        return dx, dq, gW, gB

    # ------------------------------------------------------------------------------------------------ per-layer path
    @staticmethod
    def _backward_layers(ctx, dxg, x, q):
        """Backward of the per-layer path: pair-sum broadcast + last ReLU gate, then per layer {wgrad, dgrad + gate}; layer 0 through
        the pair-axis reductions (and, with the question at layer 0 and <= 32 features per object, its weight gradient too)."""
        plan, code = ctx.plan, ctx.code
        B, n, k, Q, M, G = ctx.dims[:6]
        L, dev = plan.L, x.device
        f32 = dict(dtype=torch.float32, device=dev)
        dt = H.torch_dtype(code)
        inputs, wbwd, g_w = ctx.inputs, ctx.wbwd, ctx.g_w
        gW, gB = [None] * L, [None] * L
        dZ = torch.empty(M, G, dtype=dt, device=dev)
        H.pair_sum_bwd(dxg, ctx.HL, G, dZ, G, code, B, n * n, G)
        alg0 = alg0_wgrad_ok(plan, k)
        dq = dx = None
        for l in reversed(range(L)):
            N, A_l = plan.widths[l], inputs[l]
            kt, kp = plan.ktrue[l], plan.kpad[l]
            if not (l == 0 and alg0):
                gW[l] = grad_out(ctx.param_refs[l], (N, kt))
                gB[l] = grad_out(ctx.param_refs[L + l], (N,))
                H.g_linear_bwd_wgrad(dZ, N, A_l, kp, gW[l], gB[l], ctx.gcode, M, N, kp, kt)
            wl = g_w[l] if g_w[l].is_contiguous() else g_w[l].contiguous()
            if l == plan.inject:
                Rq = torch.empty(B, N, **f32)
                if l == 0:
                    Rj = torch.empty(B * n, N, **f32); Ri = torch.empty(B * n, N, **f32)
                    H.pair_reduce_bwd(dZ, N, Rj, Ri, Rq, code, B, n, N)
                else:
                    H.pair_reduce_bwd(dZ, N, None, None, Rq, code, B, n, N)
                dq = torch.empty(B, Q, **f32)
                H.gemm_f32(Rq, N, 1, wl, kt, 1, dq, Q, B, Q, N, b_off=kt - Q)   # Rq @ W[:, -Q:]
            elif l == 0:
                Rj = torch.empty(B * n, N, **f32); Ri = torch.empty(B * n, N, **f32)
                Rq = None
                H.pair_reduce_bwd(dZ, N, Rj, Ri, None, code, B, n, N)
            if l == 0:
                if alg0:
                    gW[0] = grad_out(ctx.param_refs[0], (N, kt))
                    gB[0] = grad_out(ctx.param_refs[L], (N,))
                    H.wgrad0_from_reductions(Rj, Ri, Rq, x, q, gW[0], gB[0])
                dx = torch.empty(B, n, k, **f32)
                H.gemm_f32(Rj, N, 1, wl, kt, 1, dx, k, B * n, k, N)                        # Rj @ W0[:, 0:k]
                H.gemm_f32(Ri, N, 1, wl, kt, 1, dx, k, B * n, k, N, b_off=k, flags=H.RN_ACCUMULATE)   # + Ri @ W0[:, k:2k]
            else:
                gp = plan.widths[l - 1]
                dZp = torch.empty(M, gp, dtype=dt, device=dev)
                H.g_linear_bwd_dgrad(dZ, N, wbwd[l], N, A_l, kp, dZp, gp, ctx.gcode, M, N, gp)
                dZ = dZp
            inputs[l] = None
        return dx, dq, gW, gB


def relational_forward(x, q, mask, plan, packed, precision, g_w, g_b, f_w, f_b, label=None, coord=None):
    """-> log-probs, or (log-probs, mean NLL) when `label` is given."""
    if precision not in _INTERNAL_PRECISIONS:
        raise ValueError("precision must be one of %r" % (PRECISIONS,))
    return RelationalFunction.apply(x, q, mask, plan, packed, (precision, torch.is_grad_enabled()), label, coord, *g_w, *g_b, *f_w, *f_b)


def grid_path_ok(plan: LayerPlan, precision, B, n, k):
    """Shapes / modes whose kernels take the conv grid + the coordinate table directly: the chain path."""
    return precision in ("f16s", "f16s2") and chain_ok(plan, B, n)


def _direct_conv_ok(inp, conv_w, stride, padding):
    return (inp.dtype == torch.float32 and tuple(conv_w.shape[2:]) == (3, 3)
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
        if ctx.direct and not ctx.needs_input_grad[0]:
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
            # captured step started 10-18 us late -- and the weight gradients share a side stream that exists already instead of a
            # stream of their own: the ROCm graph executor maps the capture's streams onto 4 hardware queues, and a fourth side
            # stream ended up behind another one's kernels (the LSTM backward then started 100 us late).  Together -2.5 % on the step.
            # Which one (round 4): the layer-0 weight gradient's (idle by now), not the g_theta weight gradient's -- behind that
            # 180-us launch the first of these started ~55 us after its operands were ready.
            main, side = torch.cuda.current_stream(), _side_stream(dx.device, SCHED["conv_wgrad_stream"])
            ev = main.record_event()
            if ctx.direct and inp.shape[1] == 24:
                din = torch.empty_like(inp)
                H.conv3x3s2_bwd_data(dx, conv_w.detach().contiguous(), din)
            else:
                din = conv_bwd([True, False, False])[0]
            side.wait_event(ev)
            with torch.cuda.stream(side):
                if ctx.direct:
                    dw = grad_out(ctx.w_ref)
                    H.conv3x3s2_bwd_weight(inp, dx, dw)                # fp32 matrix pipe, no layout transposes
                else:
                    dw = conv_bwd([False, True, False])[1]
            keep = [dx, inp]
            for t in keep:
                t.record_stream(side)
            dw.record_stream(main)               # allocated on the side stream, handed to autograd on the main one

            def _join():
                cur = torch.cuda.current_stream()
                cur.wait_stream(side)
                keep.clear()
                #  a question gradient handed over by event that its consumer never took -- the gradient reached the
                # question encoder's backward as ANOTHER tensor (a second user of the question: the engine accumulated; a tensor
                # hook; retain_grad) and was read there without the wait: say so, and make this stream wait at least
                if _GRAD_EVENTS:
                    import warnings
                    for ev, _fill in _GRAD_EVENTS.values():
                        cur.wait_event(ev)
                    _GRAD_EVENTS.clear()
                    warnings.warn("the question gradient of the injected layer was handed over by event, but its consumer did not take it "
                                  "(the question has another user, a hook or retain_grad): it may have been read before it was complete -- "
                                  "set functional.SCHED['dq_async'] = 0 for such models")
            torch.autograd.Variable._execution_engine.queue_callback(_join)
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
            # (a tensor of its own: a clone is a memcpy NODE of the captured step, on the question encoder's stream, in front of
            #  the join with the conv stack; SCHED["lstm_hn_view"] = 1 returns the view instead -- A/B'd in round 6)
            return hs[T] if SCHED.get("lstm_hn_view") else hs[T].clone()
        hn = torch.empty(B, Hh, **f32)
        H.lstm_fwd(idx, *ws, None, None, None, hn)
        return hn

    @staticmethod
    def backward(ctx, dhn):
        idx, xs, gates, cs, hs, W_ih, W_hh = ctx.saved_tensors
        handed = _GRAD_EVENTS.pop(dhn.data_ptr(), None)           # a gradient that is not complete on the producing node's stream:
        if handed is not None:                                     # wait for its event; then, if given, run what fills it in -- HERE
            torch.cuda.current_stream().wait_event(handed[0])
            if handed[1] is not None:
                handed[1]()
            dhn.record_stream(torch.cuda.current_stream())
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
        if ctx.needs_input_grad[1]:
            dx = dg.mm(W_ih)
            demb = grad_out(emb_r, (ctx.vocab, xs.shape[2]))
        H.lstm_bwd_tail(idx, dx, demb, dgates, db, db2)
        dW_hh = torch.mm(dg.t(), hs[:T].reshape(T * B, -1), out=grad_out(whh_r))
        dW_ih = torch.mm(dg.t(), xs.view(T * B, -1), out=grad_out(wih_r))
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
    if logp.is_cuda and logp.dim() == 2:
        return NllMeanFunction.apply(logp, label)
    return torch.nn.functional.nll_loss(logp, label)
