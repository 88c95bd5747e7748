"""Drop-in `model.py` for mesnico/RelationNetworks-CLEVR on AMD MI355X (gfx950).

Same public surface as the reference module (`/root/reference/model.py`):
class names, constructor / forward signatures, attribute names and state_dict
keys, so `from model import RN` in a train.py-style
driver keeps working and both released checkpoints load strictly.  What is
different is everything under `RelationalLayer.forward`: the repeat / cat /
Linear / relu / sum op sequence of model.py:104-162 is replaced by hand-written
HIP kernels reached through the C-ABI library librn_hip.so (functional.py).

The conv stack and the question encoder keep the reference's torch.nn modules as parameter holders (checkpoint keys); on the GPU their
forward / backward run through this library's kernels too (rn_conv.hip, rn_convnorm.hip, rn_lstm.hip: the 3x3 / stride-2 / 24-channel
convolutions with their batch norms, the one-layer LSTM) -- outside the SURVEY section-8 hot path (section 2 #4, #5), built because they
are half of the step's time; other shapes fall back to MIOpen / torch, and the LSTM's two weight-gradient products are torch.mm.

Deliberate deviations from reference quirks:
  C1  the coordinate tensor is cached per (batch, grid, device), every entry kept for the life of
      the module, instead of "first batch size forever" -- a different later batch size works
      instead of raising, and a captured hipGraph's entry is never freed under it.
  C2  .cuda() returns self (the reference returns None; callers ignore the value).
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn
import torch.nn.functional as F

try:                                    # imported as part of the package ...
    from . import functional as RF
    from . import rn_hip as H
    from .options import OPT
except ImportError:                     # ... or flat, exactly like the reference: `from model import RN`
    import importlib.util
    import sys

    _here = os.path.dirname(os.path.abspath(__file__))
    _spec = importlib.util.spec_from_file_location("relationnetworks_clevr_amd", os.path.join(_here, "__init__.py"),
                                                   submodule_search_locations=[_here])
    _pkg = importlib.util.module_from_spec(_spec)
    sys.modules.setdefault("relationnetworks_clevr_amd", _pkg)
    from relationnetworks_clevr_amd import functional as RF          # type: ignore
    from relationnetworks_clevr_amd import rn_hip as H               # type: ignore
    from relationnetworks_clevr_amd.options import OPT               # type: ignore


class ConvInputModel(nn.Module):
    """4 x [3x3 stride-2 conv (24 ch) -> BatchNorm -> ReLU]: 128x128 image -> 8x8x24 grid
    (reference model.py:9-36; parameter names conv1..4 / batchNorm1..4 are part of the
    checkpoint contract)."""

    def __init__(self):
        super().__init__()
        cin = 3
        for i in range(1, 5):
            self.add_module("conv%d" % i, nn.Conv2d(cin, 24, 3, stride=2, padding=1))
            self.add_module("batchNorm%d" % i, nn.BatchNorm2d(24))
            cin = 24

    def forward(self, img):
        x = img
        fused = (img.is_cuda and img.dtype == torch.float32
                 and not (torch.is_grad_enabled() and not self.training))
        for i in range(1, 5):
            conv, bn = self._modules["conv%d" % i], self._modules["batchNorm%d" % i]
            hw = ((x.shape[2] + 2 * conv.padding[0] - 3) // conv.stride[0] + 1) * ((x.shape[3] + 2 * conv.padding[1] - 3) // conv.stride[1] + 1)
            if fused and hw % 4 == 0 and bn.track_running_stats and bn.momentum is not None:
                # the package's own stride-2 3x3 convolution (rn_conv.hip) + the fused batch-norm / ReLU kernels (rn_convnorm.hip)
                x = RF.ConvBNReLUFunction.apply(x, conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                                bn.num_batches_tracked, self.training, bn.momentum, bn.eps, conv.stride, conv.padding)
            else:
                x = F.relu(bn(conv(x)))
        return x


class QuestionEmbedModel(nn.Module):
    """Embedding(in_size+1, embed) -> 1-layer LSTM(batch_first) -> final hidden state (B, hidden)
    (reference model.py:39-58)."""

    def __init__(self, in_size, embed=32, hidden=128):
        super().__init__()
        self.wembedding = nn.Embedding(in_size + 1, embed)
        self.lstm = nn.LSTM(embed, hidden, batch_first=True)
        self.hidden = hidden

    def forward(self, question):
        l = self.lstm
        if (question.is_cuda and l.num_layers == 1 and not l.bidirectional
                and l.batch_first and l.bias and getattr(l, "proj_size", 0) == 0 and l.input_size == 32 and l.hidden_size == 128
                and self.wembedding.padding_idx is None and self.wembedding.max_norm is None and l.weight_ih_l0.dtype == torch.float32):
            # embedding + the whole recurrence in one launch per direction (rn_lstm.hip)
            return RF.QuestionLSTMFunction.apply(question, self.wembedding.weight, l.weight_ih_l0, l.weight_hh_l0, l.bias_ih_l0, l.bias_hh_l0)
        _, (h_n, _c_n) = self.lstm(self.wembedding(question))
        return h_n[0]


class RelationalLayerBase(nn.Module):
    """Owns f_phi (f_fc1/2/3 + dropout) -- reference model.py:60-78."""

    def __init__(self, in_size, out_size, qst_size, hyp):
        super().__init__()
        self.f_fc1 = nn.Linear(hyp["g_layers"][-1], hyp["f_fc1"])
        self.f_fc2 = nn.Linear(hyp["f_fc1"], hyp["f_fc2"])
        self.f_fc3 = nn.Linear(hyp["f_fc2"], out_size)
        self.dropout = nn.Dropout(p=hyp["dropout"])
        # draw counter of the library's dropout-mask generator (rn_dropout_mask): {draws so far, 0}.  A NON-persistent buffer: it moves
        # with .cuda(), the trainer's copy guard puts it back like BatchNorm's buffers, and state_dict() keeps the reference's keys
        self.register_buffer("_dropout_draws", torch.zeros(2, dtype=torch.int64), persistent=False)
        self._dropout_seed = None                               # drawn from torch's CPU generator at the first mask (torch.manual_seed governs it)
        self.on_gpu = False
        self.hyp = hyp
        self.qst_size = qst_size
        self.in_size = in_size
        self.out_size = out_size

    def cuda(self, device=None):
        self.on_gpu = True
        return super().cuda(device)


class RelationalLayer(RelationalLayerBase):
    """g_theta over all n^2 object pairs -> sum -> f_phi (reference model.py:81-162), on HIP kernels.

    `g_layers` stays an indexable nn.ModuleList of nn.Linear (the checkpoint keys
    rl.g_layers.{i}.{weight,bias} and extract.py's forward hooks depend on it); the Linear
    modules only hold the parameters -- their own forward is never used on the hot path."""

    def __init__(self, in_size, out_size, qst_size, hyp, extraction=False):
        super().__init__(in_size, out_size, qst_size, hyp)
        self.quest_inject_position = hyp["question_injection_position"]
        self.in_size = in_size
        self.g_layers_size = hyp["g_layers"]
        layers = []
        for idx, width in enumerate(self.g_layers_size):
            fan_in = in_size if idx == 0 else self.g_layers_size[idx - 1]
            if idx == self.quest_inject_position:
                fan_in += qst_size                       # the "h" layer that also sees the question
            layers.append(nn.Linear(fan_in, width))
        self.g_layers = nn.ModuleList(layers)
        self.extraction = extraction
        # MI355X execution options
        # "auto" -> "f16s" where a kernel for it covers the shape (log-probs within ~2e-4 of the fp32 reference), else "fp32"
        # (exact-fp32 MFMA, ~5e-7): whatever "auto" picks meets the 1e-3 bar.  "bf16" (~1e-2) only on request.  DESIGN.md section 2.
        self.precision = hyp.get("precision", OPT.precision)
        # eval(): the chain path with hi + lo split weights on EVERY g layer (no tile-dithered images) whenever nothing needs a gradient
        # -- a question's log-probs then do not depend on its position in the batch or on the order of its objects (train.py:98-105
        # evaluates with whatever batch size fits).  False: eval() runs the training arithmetic.  INTEGRATION.md section 1.
        self.eval_two_pass = bool(hyp.get("eval_two_pass", OPT.eval_two_pass))
        self.forced_dropout_mask = None                  # tests: explicit (B, f_fc2) mask incl. 1/(1-p)
        self._packed = RF.PackedWeights()
        self._plan_cache = {}

    # -- helpers ---------------------------------------------------------------------------
    def _plan(self, k):
        key = (k, self.qst_size)
        if key not in self._plan_cache:
            if 2 * k != self.in_size:
                raise RuntimeError("object size %d does not match rl_in_size %d" % (k, self.in_size))
            self._plan_cache[key] = RF.LayerPlan(k, self.qst_size, self.g_layers_size, self.quest_inject_position)
        return self._plan_cache[key]

    def draw_dropout_ahead(self, b, device):
        """Draw this forward pass's dropout mask NOW (RN.forward: on the question encoder's side stream, instead of three
        small launches between the conv stack and the g chain).  The mask is the pass's only RNG draw either way."""
        self._mask_ahead = None
        if self.forced_dropout_mask is None and self.training and self.dropout.p > 0 and not self._hooked():
            self._mask_ahead = (b, self._dropout_mask(b, device))

    def _dropout_mask(self, b, device):
        ahead, self._mask_ahead = getattr(self, "_mask_ahead", None), None
        if ahead is not None and ahead[0] == b and ahead[1].device == device:
            return ahead[1]
        if self.forced_dropout_mask is not None:
            return self.forced_dropout_mask.to(device=device, dtype=torch.float32)
        if self.training and self.dropout.p > 0:
            dev = torch.device(device)
            if OPT.native_dropout and dev.type == "cuda" and self.dropout.p < 1 and self._dropout_draws.device == dev:
                # the library's own counter-based generator, advanced on the device by the launch itself: a captured step that draws
                # from torch's generator makes every replay launch two fill kernels (Philox seed / offset) in front of the graph
                if self._dropout_seed is None:
                    import torch.distributed as dist
                    rank = dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0
                    # (every rank draws its own masks: identically seeded replicas would otherwise drop the same units of their shards)
                    self._dropout_seed = (int(torch.randint(0, 2 ** 62, (1,)).item()) + rank * 0x9E3779B97F4A7C15) & (2 ** 63 - 1)
                return H.dropout_mask(torch.empty(b, self.f_fc2.out_features, device=dev), self.dropout.p, self._dropout_seed, self._dropout_draws)
            # same RNG consumption as the reference's self.dropout(x_f) on a (B, f_fc2) tensor
            return self.dropout(torch.ones(b, self.f_fc2.out_features, device=device))
        return None

    def _hooked(self):
        return self.extraction or any(len(l._forward_hooks) for l in self.g_layers)

    def grid_fast_path(self, b, n, k):
        """True when RN.forward may hand over the conv grid and the coordinate table separately (coordinate tagging inside
        the kernels, model.py:195-201) instead of the concatenated (B, n, k) objects."""
        if self._hooked() or 2 * k != self.in_size:
            return False
        return RF.grid_path_ok(self._plan(k), self.resolved_precision(b, n, k), b, n, k)

    def forward(self, x, qst, label=None, coord=None):
        """label (int64 (B,), optional, not part of the reference signature): also return the mean NLL of train.py:41,
        computed inside the f_phi launches -> (log_probs, loss).  coord ((2, n) fp32, optional, only with grid_fast_path):
        x then holds the conv features only, (B, n, k - 2)."""
        b, d, k = x.size()
        if coord is not None:
            k += coord.shape[0]
        plan = self._plan(k)
        if self._hooked():
            self._packed.q_grad_async = False                   # (the permission is ONE relational_forward call's; this path makes none)
            out = self._forward_hook_compat(x, qst, plan)
            return out if label is None else (out, RF.nll_loss_mean(out, label))
        g_w = [l.weight for l in self.g_layers]
        g_b = [l.bias for l in self.g_layers]
        f_w = [self.f_fc1.weight, self.f_fc2.weight, self.f_fc3.weight]
        f_b = [self.f_fc1.bias, self.f_fc2.bias, self.f_fc3.bias]
        mask = self._dropout_mask(b, x.device)
        prec = self.resolved_precision(b, d, k)
        if prec == "f16s" and not self.training and self.eval_two_pass:
            prec = "f16s2"                                      # (functional: two passes where no gradient is needed, else plain f16s)
        return RF.relational_forward(x, qst, mask, plan, self._packed, prec, g_w, g_b, f_w, f_b, label=label, coord=coord)

    def resolved_precision(self, b, d, k):
        """The arithmetic mode a forward pass on (b, d, k) objects runs in: `self.precision`, with "auto" resolved to
        "f16s" wherever the register-resident chains cover the shape (functional.chain_ok), else "fp32" (512-wide g layers, other
        depths: the per-layer fp32-MFMA kernels) -- both meet the 1e-3 log-prob bar; "auto" never lands on single-pass bf16.
        An explicit "f16s" on a shape the chains do not cover raises; "bf16" always means the per-layer bf16 kernels."""
        if self.precision != "auto":
            return self.precision
        return "f16s" if RF.f16s_ok(self._plan(k), b, d) else "fp32"

    @torch.no_grad()
    def extract_features(self, x, qst, layer_idx):
        """R-CBIR features of extract.py:49-74 without a hook and without the matrix the hook reads: the input of g layer
        `layer_idx` -- (B n^2, in) in the reference: 402 MB for the injected layer of the IR models at B = 64 -- is formed tile by
        tile on chip (rn_extract_features: pair rows in LDS, g layers 0 .. layer_idx-1 in fp32 on the matrix pipe with the tile
        resident), L2-normalised per pair and reduced to (max, mean) over each question's pairs; the question columns of an
        injection layer are excluded (extract.py:66-67).  Returns two (B, F) fp32 tensors, F = 2k for layer 0, else the g width.
        SURVEY.md 8f row N3.  The 512-wide *-sd models take the per-layer kernels + rn_pair_features (a materialised fp32 input:
        M = B * 144 rows only); registering a forward hook keeps working for every model (_forward_hook_compat)."""
        b, d, k = x.size()
        plan = self._plan(k)
        if not 0 <= layer_idx < plan.L:
            raise ValueError("layer_idx must be in [0, %d)" % plan.L)
        H._dev(x, "x")
        Q = qst.shape[1]
        if all(w_ == 256 for w_ in plan.widths[:layer_idx]) and 2 * k <= 256 and layer_idx <= 4:
            xs, q = x.float(), qst.float().contiguous()
            Wts, biases, per_q = [], [], []
            for l in range(layer_idx):
                W, bias = self.g_layers[l].weight.detach(), self.g_layers[l].bias.detach()
                kin = 2 * k if l == 0 else plan.widths[l - 1]
                Wts.append(W[:, :kin].t().contiguous())
                if l == plan.inject:                           # [.. | q] @ W^T + b = .. @ W[:, :kin]^T + (q @ W[:, kin:]^T + b): a per-question bias row
                    biases.append(torch.addmm(bias, q, W[:, kin:].t()).contiguous())
                    per_q.append(True)
                else:
                    biases.append(bias.contiguous())
                    per_q.append(False)
            F_ = 2 * k if layer_idx == 0 else plan.widths[layer_idx - 1]
            return H.extract_features(xs, Wts, biases, per_q, F_)
        code = H.dtype_code("bf16" if self.precision == "bf16" else "fp32")     # (the per-layer kernels; parity-clean unless bf16 is asked for)
        wfwd, _ = self._packed.get(plan, [l.weight for l in self.g_layers], code, bwd_images=False)
        gb = [l.bias.detach().contiguous() for l in self.g_layers]
        inputs, _ = RF.layers_forward(x.float(), qst.float().contiguous(), plan, gb, wfwd, code, keep_inputs=True, stop_at=layer_idx)
        A = inputs[layer_idx]
        F_ = plan.ktrue[layer_idx] - (Q if layer_idx == plan.inject else 0)
        if F_ % 64:
            raise RuntimeError("feature width %d of layer %d is not a multiple of 64 (use the hook path)" % (F_, layer_idx))
        return H.pair_features(A, A.shape[1], F_, code, b, d * d)

    @torch.no_grad()
    def _forward_hook_compat(self, x, qst, plan):
        """extract.py registers forward hooks on g_layers[k] and reads the layer *input*
        (B*n*n, in) (extract.py:43,64-68).  When hooks (or extraction=True) are present the
        chain runs layer by layer on the same HIP kernels and each hooked layer's materialised
        input / output is handed to its hooks as fp32 tensors; inference only."""
        # (the per-layer kernels exist in bf16 and fp32: "auto" / "f16s" take the parity-clean one -- the hooks then see the
        # reference's fp32 activations to ~1e-6)
        code = H.dtype_code("bf16" if self.precision == "bf16" else "fp32")
        H._dev(x, "x")
        x = x.float()
        q = qst.float().contiguous()
        wfwd, _ = self._packed.get(plan, [l.weight for l in self.g_layers], code)
        gb = [l.bias.detach().contiguous() for l in self.g_layers]

        def layer_hook(l, a_in, h_out):
            layer = self.g_layers[l]
            if len(layer._forward_hooks):
                inp = a_in[:, : plan.ktrue[l]].float()
                outp = h_out[:, : plan.widths[l]].float()      # note: post-ReLU (bias+ReLU are fused)
                for hook in list(layer._forward_hooks.values()):
                    hook(layer, (inp,), outp)

        _inputs, HL = RF.layers_forward(x, q, plan, gb, wfwd, code, keep_inputs=False, layer_hook=layer_hook)
        if self.extraction:
            return None                                        # reference model.py:147-148
        B = x.shape[0]
        G = plan.widths[-1]
        xg = torch.empty(B, G, dtype=torch.float32, device=x.device)
        H.pair_sum_fwd(HL, G, xg, code, B, x.shape[1] ** 2, G)
        fw = [m.weight.detach().contiguous() for m in (self.f_fc1, self.f_fc2, self.f_fc3)]
        fb = [m.bias.detach().contiguous() for m in (self.f_fc1, self.f_fc2, self.f_fc3)]
        mask = self._dropout_mask(B, x.device)
        return RF.f_phi_forward(xg, fw, fb, mask)[2]


class RN(nn.Module):
    """conv grid / state description -> objects with coordinate tags -> LSTM question ->
    relational layer (reference model.py:164-223)."""

    def __init__(self, args, hyp, extraction=False):
        super().__init__()
        self.coord_tensor = None
        self._coord_cache = {}                 # (b, d, device) -> (b, 2, d*d); entries are never evicted (see _coords)
        self._coord_tables = {}                # (d, device) -> (2, d*d): the batch-independent table the kernels read
        self._side_stream = None
        self.overlap_streams = OPT.overlap_streams
        self.on_gpu = False
        self.conv = ConvInputModel()
        self.state_desc = hyp["state_description"]
        hidden_size = hyp["lstm_hidden"]
        self.text = QuestionEmbedModel(args.qdict_size, embed=hyp["lstm_word_emb"], hidden=hidden_size)
        self.rl_in_size = hyp["rl_in_size"]
        self.rl_out_size = args.adict_size
        self.rl = RelationalLayer(self.rl_in_size, self.rl_out_size, hidden_size, hyp, extraction)
        print("Supposing IR model" if hyp["question_injection_position"] != 0 else "Supposing original DeepMind model")

    def build_coord_tensor(self, b, d, device=None):
        """(B, 2, d, d): channel 0 = x = lin[col], channel 1 = y = lin[row], lin = linspace(-d/2, d/2, d)
        computed on the CPU exactly as the reference does (model.py:208-218) and then moved."""
        lin = torch.linspace(-d / 2.0, d / 2.0, d)
        ct = torch.stack((lin.unsqueeze(0).expand(d, d), lin.unsqueeze(1).expand(d, d)))
        ct = ct.unsqueeze(0).expand(b, 2, d, d).contiguous()
        if device is not None:
            ct = ct.to(device)
        elif self.on_gpu:
            ct = ct.cuda()
        self.coord_tensor = ct
        return ct

    def _coords(self, b, d, device):
        """The (b, 2, d*d) coordinate channels, one tensor per (batch, grid, device) for the life of the module: a
        captured hipGraph bakes the pointer in, so an entry must never be freed while the module lives (an eval batch of
        another size between two replays allocates ITS OWN entry instead of replacing the training one)."""
        key = (b, d, device)
        ct = self._coord_cache.get(key)
        if ct is None:
            ct = self._coord_cache[key] = self.build_coord_tensor(b, d, device).view(b, 2, d * d)
        self.coord_tensor = ct                 # the reference's attribute: the tensor of the latest forward
        return ct

    def _coord_table(self, d, device):
        """(2, d*d) fp32: row 0 = x = lin[p % d], row 1 = y = lin[p // d] (build_coord_tensor for one sample), kept for the
        life of the module (a captured hipGraph reads it)."""
        key = (d, device)
        ct = self._coord_tables.get(key)
        if ct is None:
            keep = self.coord_tensor
            ct = self._coord_tables[key] = self.build_coord_tensor(1, d, device).view(2, d * d).contiguous()
            self.coord_tensor = keep
        return ct

    def _text_on_side_stream(self, qst_idxs, after=None):
        """The LSTM is a serial chain of ~20 tiny kernels that leaves the chip empty; fork it onto a
        second HIP stream so it (and, through autograd's stream bookkeeping, its backward) overlaps
        the conv stack.  Joined before the relational layer; capturable in a hipGraph."""
        cur = torch.cuda.current_stream()
        if self._side_stream is None:
            self._side_stream = RF._side_stream(qst_idxs.device, "text")    # one per process and device: roles never alias (RF.fresh_stream)
        side = self._side_stream
        if after is not None:
            side.wait_event(after)
        else:
            side.wait_stream(cur)
        with torch.cuda.stream(side):
            self.rl._packed.repack_ahead()                      # this step's weight images: off the critical path, same fork/join
            self.rl.draw_dropout_ahead(qst_idxs.shape[0], qst_idxs.device)
            qst = self.text(qst_idxs)
        return qst, side

    def forward_loss(self, img, qst_idxs, label):
        """forward + F.nll_loss(output, label) (train.py:40-41) with the loss folded into the f_phi kernels: -> (log_probs, loss)."""
        return self.forward(img, qst_idxs, label=label)

    def forward(self, img, qst_idxs, label=None):
        side = None
        coord = None
        fork_ev = None
        if self.overlap_streams and qst_idxs.is_cuda and not self.state_desc:
            # the question encoder forks off HERE (it depends on nothing of this pass) but is launched -- captured -- behind the conv
            # stack: with the fork in front, the replayed graph's first conv kernels started late (-1.8 % on the step, same box;
            # DESIGN.md, "capture order")
            fork_ev = torch.cuda.current_stream().record_event()
        if self.state_desc:
            x = img                                             # (B, 12, 7) state descriptions
        else:
            if fork_ev is not None and RF.SCHED.get("text_first"):      # (A/B knob: the question encoder captured IN FRONT of the conv stack)
                qst, side = self._text_on_side_stream(qst_idxs, after=fork_ev)
                fork_ev = None
            x = self.conv(img)                                  # (B, 24, d, d)
            if fork_ev is not None:
                qst, side = self._text_on_side_stream(qst_idxs, after=fork_ev)
            b, k, d, _ = x.size()
            coord = None
            if self.rl.grid_fast_path(b, d * d, k + 2):
                # no concatenation: the relation kernels read the grid through this strided view and tag the coordinates
                # themselves; the input gradient comes back in the grid's layout
                coord = self._coord_table(d, x.device)
                x = x.view(b, k, d * d).permute(0, 2, 1)        # (B, d*d, 24)
            else:
                x = torch.cat([x.view(b, k, d * d), self._coords(b, d, x.device)], 1).permute(0, 2, 1)   # (B, d*d, 26) strided view
        if side is None:
            qst = self.text(qst_idxs)
            self.rl._packed.q_grad_async = False
        else:
            cur = torch.cuda.current_stream()
            cur.wait_stream(side)
            qst.record_stream(cur)
            # the question's gradient goes to the question encoder's backward on ITS stream and to nothing else: a relational layer
            # that produces it on a side stream of its own (question injected behind layer 0) may hand it over by event
            self.rl._packed.q_grad_async = type(qst.grad_fn).__name__ == "QuestionLSTMFunctionBackward"
            ahead = getattr(self.rl, "_mask_ahead", None)
            if ahead is not None:
                ahead[1].record_stream(cur)
        if not self.state_desc and coord is not None:
            return self.rl(x, qst, label=label, coord=coord)
        return self.rl(x, qst) if label is None else self.rl(x, qst, label=label)

    @torch.no_grad()
    def extract_features(self, img, qst_idxs, layer_idx):
        """extract.py's R-CBIR features (max / mean over pairs of the L2-normalised input of g layer `layer_idx`) through the
        native op instead of a forward hook: RelationalLayer.extract_features."""
        if self.state_desc:
            x = img
        else:
            x = self.conv(img)
            b, k, d, _ = x.size()
            x = torch.cat([x.view(b, k, d * d), self._coords(b, d, x.device)], 1).permute(0, 2, 1)
        return self.rl.extract_features(x, self.text(qst_idxs), layer_idx)

    def cuda(self, device=None):
        self.on_gpu = True
        self.rl.on_gpu = True
        return super().cuda(device)
