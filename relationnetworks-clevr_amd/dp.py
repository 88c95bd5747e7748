"""Data-parallel training over the GPUs of one node: one process per GPU, ONE flat fp32 gradient
bucket, ONE RCCL all-reduce per step over xGMI.

Replaces the reference's single-process `torch.nn.DataParallel` (train.py:256-258), which
re-broadcasts the parameters and reduce-adds the gradients onto GPU 0 every step.  Here the
replicas stay resident; per step each rank computes the mean-loss gradient of its own 1/world
shard of the global minibatch, the buckets are summed with all-reduce and scaled by 1/world,
which equals the gradient of the mean loss over the global batch (train.py:41 + DataParallel's
gather).  BatchNorm statistics stay per replica, exactly like DataParallel (no SyncBN).

The whole model has ~485k parameters (1.94 MB of fp32 gradients for *-fp): a single
latency-bound collective, so everything goes into one bucket and nothing is overlapped with
backward (the bucket is complete only when backward ends: the g_theta weight gradients, the
bulk of the bytes, are produced last-layer-first but conv/LSTM grads arrive at the very end)."""
from __future__ import annotations

import contextlib
import datetime
import os
import sys
import threading
import warnings

import torch
import torch.distributed as dist

try:
    from . import functional as RF
    from .options import OPT
except ImportError:                                   # imported through the top-level shim
    from relationnetworks_clevr_amd import functional as RF          # type: ignore
    from relationnetworks_clevr_amd.options import OPT               # type: ignore


def _immortal(obj):
    """Keep `obj` alive for the life of the process, interpreter shutdown included.  For a torch.cuda.CUDAGraph whose capture
    FAILED: its destructor calls into the runtime with the invalidated capture and throws -- from a destructor, i.e.
    std::terminate (seen with a gloo collective inside a capture).  A leaked handle costs nothing; an abort costs the job."""
    import ctypes
    if obj is not None:
        ctypes.pythonapi.Py_IncRef(ctypes.py_object(obj))


class Watchdog:
    """A deadline on a region that waits for the GPU or for other ranks.  A collective whose peers never arrive (a rank that
    fell back to another mode, a hung capture) blocks for ever and says nothing; under `guard(what, seconds)` the process
    instead prints what it was waiting for and EXITS (code 124, like timeout(1)) -- torch.distributed.run then tears the job
    down.  seconds <= 0: no deadline."""

    EXIT_CODE = 124
    on_expire = None          # optional callable(what, seconds), run before the exit (bench.py: its one JSON line, marked as a failure)

    def __init__(self, rank=0):
        self.rank = rank

    def _expire(self, what, seconds):
        if Watchdog.on_expire is not None:
            try:
                Watchdog.on_expire(what, seconds)
            except Exception:
                pass
        sys.stderr.write("[relationnetworks_clevr_amd] rank %d: no progress for %.0f s in: %s -- giving up (exit %d).  "
                         "If this is the in-graph gradient all-reduce, re-run with RN_NO_GRAPH_ALLREDUCE=1.\n"
                         % (self.rank, seconds, what, self.EXIT_CODE))
        sys.stderr.flush()
        os._exit(self.EXIT_CODE)

    @contextlib.contextmanager
    def guard(self, what, seconds):
        if not seconds or seconds <= 0:
            yield
            return
        t = threading.Timer(seconds, self._expire, args=(what, seconds))
        t.daemon = True
        t.start()
        try:
            yield
        finally:
            t.cancel()


class ControlPlane:
    """Rank agreement that depends on neither the GPU nor the data communicator: a CPU-side `gloo` group over the same ranks
    (the group itself when it already is gloo).  The trainer decides HOW a step runs (all-reduce inside the captured graph or
    eager behind it) from things that can differ per rank -- a capture that throws, a self-check that mismatches -- and a
    decision taken per rank leaves the ranks in different modes: mismatched collectives, a hang.  Every such decision goes
    through all_ok() / any_of().  If no gloo group can be made the flags travel over `group` as device tensors."""

    def __init__(self, group=None, timeout_s=300.0, device=None):
        self.group, self.device = group, device
        self.world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        self.rank = dist.get_rank(group) if self.world > 1 else 0
        self.cpu_group = None
        self._cpu = False
        if self.world > 1:
            if dist.get_backend(group) == "gloo":
                self.cpu_group, self._cpu = group, True
            else:
                try:
                    ranks = dist.get_process_group_ranks(group) if group is not None else None
                    self.cpu_group = dist.new_group(ranks=ranks, backend="gloo", timeout=datetime.timedelta(seconds=max(timeout_s, 10.0)))
                    self._cpu = True
                except Exception as e:                        # a build without gloo: the data communicator carries the flags
                    warnings.warn("no gloo control group (%s): rank agreement goes over the data communicator" % str(e)[:120])

    def _reduce(self, value, op):
        if self.world == 1:
            return value
        if self._cpu:
            t = torch.tensor([float(value)], dtype=torch.float64)
            dist.all_reduce(t, op=op, group=self.cpu_group)
        else:
            t = torch.tensor([float(value)], dtype=torch.float64, device=self.device)
            dist.all_reduce(t, op=op, group=self.group)
        return float(t.item())

    def all_ok(self, ok) -> bool:
        """True iff `ok` on EVERY rank."""
        return self._reduce(1.0 if ok else 0.0, dist.ReduceOp.MIN) > 0.5

    def any_of(self, flag) -> bool:
        return self._reduce(1.0 if flag else 0.0, dist.ReduceOp.MAX) > 0.5

    def gather(self, obj):
        """[obj of rank 0, obj of rank 1, ...] on every rank (small python objects: flags, checksums, reasons)."""
        if self.world == 1:
            return [obj]
        out = [None] * self.world
        if self._cpu:
            dist.all_gather_object(out, obj, group=self.cpu_group)
        else:
            dist.all_gather_object(out, obj, group=self.group)
        return out

    def ranks_seen(self):
        return sorted(self.gather(self.rank))


class FlatGradBucket:
    """One contiguous fp32 buffer holding every parameter's gradient (the all-reduce message).

    Two ways to fill it:
      * accumulate mode (`zero_()` then backward): every p.grad is a view into the buffer, autograd
        adds into it in place -- one add kernel per parameter;
      * gather mode (`detach_()`, backward, `gather_()`): backward produces fresh gradient tensors
        (no accumulation kernels, no memset) and ONE concatenation kernel packs them into the buffer,
        after which the views are re-attached for the optimizer.  Used by the trainer."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        dev = self.params[0].device
        offs, total = [], 0
        for p in self.params:
            if p.dtype != torch.float32:
                raise TypeError("FlatGradBucket expects fp32 master parameters")
            total = (total + 3) // 4 * 4                  # every slot starts on 16 bytes: backward kernels write dW / db straight
            offs.append(total)                            # into the views with vector stores (the pad floats stay zero)
            total += p.numel()
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.numel = total
        self.offsets = offs
        self.views = [self.flat[o:o + p.numel()].view_as(p) for p, o in zip(self.params, offs)]
        self._zeros = {}
        self._known_zero = set()                          # slots of parameters without a gradient that are already zero
        RF.register_grad_slots(self.params, self.flat, offs)     # backward kernels may then write gradients in place (gather mode)
        self.attach_()

    def attach_(self):
        for p, v in zip(self.params, self.views):
            p.grad = v

    def detach_(self):
        """Before backward in gather mode: autograd then assigns instead of accumulating."""
        for p in self.params:
            p.grad = None
        # a backward pass that raised part-way never ran the engine callback that ends its slot hand-outs: start clean
        RF.reset_grad_slot_handouts()

    def gather_(self):
        """Make the flat buffer hold this backward pass's gradients and re-attach the views.  Gradients the backward kernels
        wrote straight into their slots (functional.grad_out) are already in place; if every one is, nothing is launched;
        if none is, ONE concatenation kernel packs them; a few strays are copied one by one."""
        base, stray = self.flat.data_ptr(), []
        for p, o in zip(self.params, self.offsets):
            g = p.grad
            if g is not None and g.is_contiguous() and g.data_ptr() == base + 4 * o:
                self._known_zero.discard(o)                 # a backward kernel wrote the slot in place: it is no longer known to be zero
                continue
            stray.append((p, o, g))
        if len(stray) == len(self.params):
            parts = []
            for p, o, g in stray:
                if g is None:                               # parameter unused in this graph (e.g. conv for *-sd)
                    g = self._zeros.get(p.numel())
                    if g is None:
                        g = self._zeros[p.numel()] = torch.zeros(p.numel(), dtype=torch.float32, device=self.flat.device)
                parts.append(g.reshape(-1))
            if self.numel == sum(p.numel() for p in self.params):
                torch.cat(parts, out=self.flat)
            else:                                           # (padded slots: no single concatenation)
                for (p, o, _g), part in zip(stray, parts):
                    self.flat[o:o + p.numel()].copy_(part)
            self._known_zero.clear()
        else:
            # nothing but this method writes the slot of a parameter that has no gradient (the optimiser reads it): zeroed
            # once, it stays zero -- no fill launch per unused parameter and step (the 16 conv / BN tensors of the *-sd models)
            for p, o, g in stray:
                if g is None:
                    if o not in self._known_zero:
                        self.flat[o:o + p.numel()].zero_()
                        self._known_zero.add(o)
                else:
                    self._known_zero.discard(o)
                    self.flat[o:o + p.numel()].copy_(g.reshape(-1))
        self.attach_()

    def zero_(self):
        """Accumulate mode: replaces optimizer.zero_grad(): one memset, the .grad views stay attached."""
        self.flat.zero_()
        self._known_zero.clear()

    def check_attached(self):
        for p in self.params:
            g = p.grad
            if g is None or g.untyped_storage().data_ptr() != self.flat.untyped_storage().data_ptr():
                raise RuntimeError("a .grad left the flat bucket (zero_grad(set_to_none=True) was called?)")

    def all_reduce_mean(self, group=None, scale=True, even_alone=False):
        """sum over ranks, then scale by 1/world -> gradient of the global-batch mean loss.  scale=False leaves the
        1/world factor to the caller (the fused clip + Adam kernel applies it); returns that factor.  even_alone: launch the
        collective on a one-rank communicator too (DataParallelTrainer(single_rank_exchange=True))."""
        if dist.is_available() and dist.is_initialized():
            world = dist.get_world_size(group)
            if world > 1 or even_alone:
                dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
                if scale:
                    self.flat.mul_(1.0 / world)
                return 1.0 / world
        return 1.0

    def clip_grad_norm_(self, max_norm: float, eps: float = 1e-6):
        """Global L2 clip (train.py:45-46, torch clip_grad_norm semantics) on the flat buffer:
        two kernels and no host synchronisation.  Returns the (device) total norm."""
        total = torch.linalg.vector_norm(self.flat, 2)
        coef = torch.clamp(max_norm / (total + eps), max=1.0)
        self.flat.mul_(coef)
        return total


def broadcast_module_state(module, src: int = 0, group=None):
    """Make every replica start from rank `src`'s parameters and buffers (BN running stats)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.data, src=src, group=group)


class FusedClipAdam:
    """clip_grad_norm_ + torch.optim.Adam (amsgrad=False, one parameter group) on the bucket's flat gradient: two HIP
    launches (rn_clip_adam_step) instead of the norm / clamp / scale / multi-tensor-Adam sequence.  Hyper-parameters
    are read from the torch optimizer's param_group at every step, so LR schedulers keep working; the moment
    buffers live here (the reference checkpoints model weights only, train.py:364)."""

    def __init__(self, bucket: "FlatGradBucket", optimizer):
        import numpy as np
        H = RF.H                                           # (the binding `functional` imported: works under the flat import too)
        self.H, self.bucket, self.opt = H, bucket, optimizer
        dev = bucket.flat.device
        ch = H.load().rn_clip_adam_chunk()
        rec = []
        for p, off in zip(bucket.params, bucket.offsets):
            n, done = p.numel(), 0
            while done < n:
                c = min(ch, n - done)
                rec.append((p.data_ptr() + 4 * done, off + done, c, 0))
                done += c
        arr = np.array(rec, dtype=np.dtype([("p", "<u8"), ("o", "<i8"), ("c", "<i4"), ("z", "<i4")]))
        self.chunks = torch.from_numpy(arr.view(np.uint8).copy()).to(dev)
        self.nchunks = len(rec)
        self.ptrs = [p.data_ptr() for p in bucket.params]
        self.m = torch.zeros_like(bucket.flat)
        self.v = torch.zeros_like(bucket.flat)
        self.ws = torch.empty(H.workspace_bytes(H.WS_CLIP_ADAM), dtype=torch.uint8, device=dev)
        self.norm = torch.zeros(1, dtype=torch.float32, device=dev)
        self.t = 0
        # in-graph mode (one GPU): every per-step scalar lives in device memory -- hyper = {grad_scale, max_norm, lr, beta1, beta2,
        # eps, weight_decay}, t_dev = update count -- so the two launches can be captured with the rest of the step
        self.hyper = torch.zeros(8, dtype=torch.float32, device=dev)
        self.t_dev = torch.zeros(1, dtype=torch.int32, device=dev)
        self._hyper_host = None
        self._t_dev_host = 0                               # the device count as the host knows it

    @staticmethod
    def supports(bucket, optimizer):
        if not OPT.fused_adam or type(optimizer) is not torch.optim.Adam:
            return False
        if len(optimizer.param_groups) != 1 or not bucket.flat.is_cuda:
            return False
        g = optimizer.param_groups[0]
        if g.get("amsgrad") or g.get("maximize") or g.get("capturable") or g.get("differentiable"):
            return False
        ids = {id(p) for p in g["params"]}
        return ids == {id(p) for p in bucket.params} and all(p.is_contiguous() for p in bucket.params)

    def sync_hyper(self, clip_norm, grad_scale=1.0):
        """Host side of the in-graph mode, before every replay: rewrite the device scalars if (and only if) a scheduler or the
        caller changed them, and bring the device update count in line with the host's (the two modes can be mixed)."""
        # BEFORE the replay: the captured kernels (forward, backward, clip + Adam) have the parameters' addresses baked in -- a
        # re-assigned parameter must raise before anything writes through a stale pointer
        self._check_storage()
        g = self.opt.param_groups[0]
        cur = (float(grad_scale), float(clip_norm or 0.0), float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]),
               float(g["weight_decay"]), 0.0)
        if cur != self._hyper_host:
            self.hyper.copy_(torch.tensor(cur, dtype=torch.float32), non_blocking=False)
            self._hyper_host = cur
        if self._t_dev_host != self.t:
            self.t_dev.fill_(self.t)
            self._t_dev_host = self.t

    def step_dev(self):
        """The two launches with device-side scalars (capturable); the caller runs sync_hyper() before and after_step_dev() after
        every execution."""
        self.H.clip_adam_step_dev(self.chunks, self.nchunks, self.bucket.flat, self.m, self.v, self.ws, self.hyper, self.t_dev, self.norm)

    def _check_storage(self):
        if [p.data_ptr() for p in self.bucket.params] != self.ptrs:
            raise RuntimeError("a parameter's storage moved since the trainer was built (load_state_dict copies in place; .data = ... does not)")

    def after_step_dev(self):
        self.t += 1
        self._t_dev_host = self.t                          # (the kernel incremented its own count)
        torch.autograd.graph.increment_version(self.bucket.params)
        return self.norm

    def step(self, clip_norm, grad_scale=1.0):
        self._check_storage()
        g = self.opt.param_groups[0]
        self.t += 1
        self.H.clip_adam_step(self.chunks, self.nchunks, self.bucket.flat, self.m, self.v, self.ws, clip_norm, g["lr"], g["betas"][0],
                              g["betas"][1], g["eps"], g["weight_decay"], self.t, self.norm, grad_scale)
        # the kernel wrote the parameters behind autograd's back: bump their version counters (the relational layer
        # keys its packed-weight cache on them)
        torch.autograd.graph.increment_version(self.bucket.params)
        return self.norm


class DataParallelTrainer:
    """model + optimizer + flat bucket: step(batch) = zero -> fwd -> nll -> bwd -> all-reduce -> clip -> Adam
    (the loop body of the reference's train(), train.py:36-48).

    use_graph=True captures the step -- ~55 kernel launches (conv/BN, the LSTM, the HIP hot path) -- into ONE hipGraph on first
    use and replays it every step.  WHAT the graph holds:
      * one rank: forward, loss, backward, clip + Adam (everything);
      * N > 1 over RCCL (backend "nccl"): the same PLUS the gradient all-reduce, so that the N > 1 step has the shape of the N = 1
        step -- but only after every rank has passed a start-up self-check of a captured all-reduce against an eager one AND
        every rank's capture of the step succeeded (ControlPlane agreement).  Otherwise, on EVERY rank alike: forward + backward
        replayed, then the all-reduce and the fused (1/world, clip, Adam) launched eagerly (`exchange_fallback` says why) -- and if
        even that graph cannot be captured on some rank (an invalidated capture poisons torch's capture state), eager steps
        everywhere (`use_graph` goes False);
      * any other backend (gloo moves the bucket through the host): the eager exchange, always.
    Inputs are copied into static buffers (or written there by the loader: input_buffers()), so shapes must not change
    between steps.  With N > 1 every wait on another rank runs under a Watchdog deadline (options.dp_timeout seconds)."""

    def __init__(self, model, optimizer, clip_norm: float | None = 50.0, group=None, use_graph: bool = False,
                 copy_guard_every: int = 512, copy_guard_max_flushed: float = 0.05, copy_guard_max_clamped: float = 1e-3,
                 graph_allreduce: bool | None = None, timeout_s: float | None = None, single_rank_exchange: bool = False):
        self.model, self.opt, self.clip_norm, self.group = model, optimizer, clip_norm, group
        # guard of the e4m3 activation copies (check_activation_copies): before the first capture / at step 1 and then every
        # `copy_guard_every` steps; 0 = never
        self.copy_guard_every, self.copy_guard_max_flushed, self.copy_guard_max_clamped = copy_guard_every, copy_guard_max_flushed, copy_guard_max_clamped
        self.copy_guard_log = []
        self._guard_done_at = None                         # value of _nstep the guard last ran for
        self._nstep = 0
        world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        self.world = world
        self.rank = dist.get_rank(group) if world > 1 else 0
        # single_rank_exchange (diagnostic): run the N > 1 machinery -- self-check, capture of the all-reduce into the step graph,
        # first-replay signature check, or the eager exchange -- on a ONE-rank communicator.  The sum over one rank is the
        # identity, so the step must equal the plain one-rank step bit for bit; what it exercises is the collective's launch,
        # capture and replay (on RCCL: the only part of the N > 1 step a one-GPU box can execute)
        self.exchange = world > 1 or (bool(single_rank_exchange) and dist.is_available() and dist.is_initialized())
        self.timeout_s = float(OPT.dp_timeout if timeout_s is None else timeout_s) if self.exchange else 0.0
        self.watchdog = Watchdog(self.rank)
        dev = next(model.parameters()).device
        self.ctl = ControlPlane(group, self.timeout_s or 300.0, dev)
        with self.watchdog.guard("broadcast of rank 0's parameters and buffers", self.timeout_s):
            broadcast_module_state(model, 0, group)
        self.bucket = FlatGradBucket(model.parameters())
        self.use_graph = use_graph
        self._graph = None
        self._static = None
        self.timing = None                                 # a list: step() appends (start, after all-reduce, after optimiser) events
        self._fused_opt = FusedClipAdam(self.bucket, optimizer) if FusedClipAdam.supports(self.bucket, optimizer) else None
        # The whole step is ONE graph whenever it can be (class docstring).  graph_allreduce: None = options.graph_allreduce and a
        # stream-ordered backend; True / False force it (tests drive the agreement logic over gloo with True)
        if graph_allreduce is None:
            graph_allreduce = OPT.graph_allreduce and self.exchange and dist.get_backend(group) == "nccl"
        coll_in_graph = not self.exchange or bool(graph_allreduce)
        self._opt_in_graph = (use_graph and self._fused_opt is not None and coll_in_graph and OPT.graph_adam)
        self.exchange_fallback = None                      # why the exchange is NOT in the graph although it was asked for
        self.exchange_checks = {}                          # what the start-up checks measured (bench.py prints them)
        self._first_graph_step_checked = False
        self._capturing = None                             # the graph object of the capture in progress / last attempted

    # ------------------------------------------------------------------------------------------- activation-copy guard
    def _h8_modules(self):
        return [m_._packed for m_ in self.model.modules() if hasattr(m_, "_packed") and hasattr(m_._packed, "h8")]

    def _h8_in_use(self):
        return OPT.h8 and any(pk.h8 for pk in self._h8_modules())

    def check_activation_copies(self, img, qst, label):
        """The e4m3 copies of H_0..2 (kept for the weight gradients of g layers 1..3) use a FIXED scale of 1: a post-ReLU value
        below 2^-10 becomes 0, one above 448 is clamped.  Fine for the released checkpoints (activations peak at 11) and for
        default-initialised models -- but nothing in the step itself would notice a model that drifts out of that range.  This
        runs ONE extra training forward on the given batch (eager, outside the step graph; BatchNorm buffers and the RNG state
        are put back) with a probe that counts, per copy, the positive activations that were flushed to zero and the bytes at the
        clamp (rn_fp8_copy_health), and switches THIS trainer's model to 16-bit copies (the relational layer's `_packed.h8 = False`
        -- not the process-wide option; the step graph is re-captured, into the same input tensors) when more than
        `copy_guard_max_flushed` of the positive activations of a layer are flushed or more than `copy_guard_max_clamped` of its
        elements are clamped.  With several ranks the decision is the OR over ranks (ControlPlane).
        -> {"layers": {l: {"positive", "flushed", "clamped", "max_value"}}, "switched": bool} or None (no e4m3 copies in use)."""
        self._guard_done_at = self._nstep
        if img.is_cuda:
            # (same cadence, same host sync: the in-launch hand-offs of the feature-split f_phi kernel have bounded spins -- a sweep
            # that gave up leaves an error word instead of a hang; training on with garbage must not be silent)
            st = RF.H.f_phi_split_status(img.device)
            any_st = bool(st)
            if self.world > 1:                                 # (raised on EVERY rank, or the peers hang in the next collective)
                with self.watchdog.guard("f_phi hand-off status: agreement over ranks", self.timeout_s):
                    any_st = self.ctl.any_of(any_st)
            if any_st:
                raise RuntimeError("rn_f_phi_split: a hand-off inside the launch was not answered (%s); results since then are "
                                   "invalid -- re-run with RN_NO_FPHI_SPLIT=1" % ("stage %d on this rank" % (st - 1) if st else "on another rank"))
        if not (self._h8_in_use() and img.is_cuda):
            return None
        bufs = [(b_, b_.clone()) for b_ in self.model.buffers()]
        rng = torch.cuda.get_rng_state(img.device)
        RF.COPY_HEALTH_PROBE = probe = []
        try:
            with torch.enable_grad():
                if hasattr(self.model, "forward_loss") and OPT.fused_loss:
                    self.model.forward_loss(img, qst, label)
                else:
                    self.model(img, qst)
        finally:
            RF.COPY_HEALTH_PROBE = None
            with torch.no_grad():
                for b_, old in bufs:
                    b_.copy_(old)
            torch.cuda.set_rng_state(rng, img.device)
        rep, bad = {}, False
        for l, t in probe:
            pos, flushed, clamped, maxb = [int(v) for v in t.cpu().tolist()]
            tot = max(pos, 1)
            # e4m3 byte -> value (positive): exponent bits 6..3 (bias 7), mantissa bits 2..0
            e, m = (maxb >> 3) & 15, maxb & 7
            maxv = (m / 8.0) * 2.0 ** -6 if e == 0 else (1.0 + m / 8.0) * 2.0 ** (e - 7)
            rep[l] = {"positive": pos, "flushed": flushed / tot, "clamped": clamped / tot, "max_value": maxv}
            bad = bad or flushed / tot > self.copy_guard_max_flushed or clamped / tot > self.copy_guard_max_clamped
        if self.world > 1:
            with self.watchdog.guard("activation-copy guard: agreement over ranks", self.timeout_s):
                bad = self.ctl.any_of(bad)
        if not probe and not bad:
            return None
        out = {"step": self._nstep, "layers": rep, "switched": bad}
        self.copy_guard_log.append(out)
        if bad:
            warnings.warn("e4m3 activation copies lose too much (%s): switching to 16-bit copies" % rep)
            for pk in self._h8_modules():
                pk.h8 = False
            self._graph = None                                 # (the captured step has the e4m3 kernels baked in)
        return out

    # --------------------------------------------------------------------------------------------------- the step's parts
    def _fwd_bwd(self, img, qst, label):
        self.bucket.detach_()
        if hasattr(self.model, "forward_loss") and img.is_cuda and OPT.fused_loss:
            _, loss = self.model.forward_loss(img, qst, label)     # F.nll_loss (mean) inside the f_phi launches
        else:
            out = self.model(img, qst)
            loss = RF.nll_loss_mean(out, label)            # F.nll_loss (mean), one launch each way on the GPU
        # the loss gradient is a cached one (loss.backward() alone launches a fill kernel for it: one more node on the
        # critical path of the captured step, between the f_phi launches)
        loss.backward(RF.unit_loss_grad(loss.device, loss.dtype) if loss.is_cuda else torch.ones_like(loss))
        self.bucket.gather_()
        return loss

    def _graph_collective(self, tensor):
        """The data-path collective as it is captured (and as the self-check captures it)."""
        dist.all_reduce(tensor, op=dist.ReduceOp.SUM, group=self.group)

    def _exchange_self_check(self):
        """Start-up check of the in-graph exchange (N > 1, before the step is captured): a bucket-sized buffer with a per-rank
        pattern goes through an EAGER all-reduce and through a CAPTURED one (a two-node graph of its own, replayed twice on
        restored inputs); the results must be bitwise equal.  -> None when that holds on EVERY rank, else the reason (a string,
        identical on every rank) -- the caller then keeps the exchange eager everywhere."""
        dev = self.bucket.flat.device
        n = self.bucket.numel
        gen = torch.Generator(device="cpu").manual_seed(1234 + self.rank)
        src = (torch.rand(n, generator=gen) - 0.5).to(dev)
        why, g, ref, buf = None, None, None, None
        try:
            ref = src.clone()
            dist.all_reduce(ref, op=dist.ReduceOp.SUM, group=self.group)     # (also the communicator's lazy initialisation)
            torch.cuda.synchronize()
            buf = src.clone()
            g = torch.cuda.CUDAGraph()
            self._capture_stream = cs = RF.fresh_stream(self.bucket.flat.device)       # (a stream of this capture's own: a failure must not poison torch's shared one)
            with torch.cuda.graph(g, stream=cs, capture_error_mode=self._capture_mode()):
                self._graph_collective(buf)
        except Exception as e:
            why = "%s: %s" % (type(e).__name__, str(e)[:160])
            self._abandon_capture(g)
        # the replay IS a collective.  A rank whose capture raised must not leave the others blocked in g.replay() on an
        # all-reduce it never joins (nobody would reach the gather below; the job would die at the watchdog instead of falling back):
        # every rank learns whether EVERY capture succeeded before ANY rank replays.
        captured = self.ctl.gather(why)
        if all(r is None for r in captured):
            try:
                for _ in range(2):
                    buf.copy_(src)
                    g.replay()
                    torch.cuda.synchronize()
                    if not torch.equal(buf, ref):
                        why = "captured all-reduce != eager all-reduce (max |diff| %.3e)" % float((buf - ref).abs().max())
                        break
            except Exception as e:
                why = "%s: %s" % (type(e).__name__, str(e)[:160])
        reasons = self.ctl.gather(why)
        self.exchange_checks["self_check"] = "passed" if all(r is None for r in reasons) else reasons
        if all(r is None for r in reasons):
            return None
        return "self-check of the captured all-reduce failed on rank(s) %s: %s" % (
            [i for i, r in enumerate(reasons) if r is not None], next(r for r in reasons if r is not None))

    def _capture(self, img, qst, label):
        # the e4m3 copy guard BEFORE the first capture (a guard that trips after it costs a second capture with its four
        # warm-up passes through the BatchNorm statistics)
        if self.copy_guard_every and self._guard_done_at is None:
            self.check_activation_copies(img, qst, label)
        if self._static is not None and all(a.shape == b.shape and a.dtype == b.dtype for a, b in zip(self._static, (img, qst, label))):
            for d_, s_ in zip(self._static, (img, qst, label)):    # a re-capture keeps the input tensors a loader may be writing into
                if d_.data_ptr() != s_.data_ptr():
                    d_.copy_(s_)
        else:
            self._static = (img.clone(), qst.clone(), label.clone())
        side = RF._side_stream(self.bucket.flat.device, "warmup")
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                   # warm-up outside capture (MIOpen find, allocator, packs)
            for _ in range(2):
                self._fwd_bwd(*self._static)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        if self._opt_in_graph and self.exchange:
            with self.watchdog.guard("start-up self-check + capture of the in-graph gradient all-reduce", self.timeout_s):
                why = self._exchange_self_check()
                graph = None
                if why is None:
                    err = None
                    try:
                        graph = self._capture_graph(True)
                    except Exception as e:              # a backend / runtime that cannot capture its collective
                        err = "%s: %s" % (type(e).__name__, str(e)[:160])
                        self._abandon_capture(self._capturing)
                    # the decision is the JOB's, not the rank's: one rank replaying an in-graph all-reduce while another launches
                    # an eager one is a hang
                    errs = self.ctl.gather(err)
                    self.exchange_checks["capture"] = "ok" if all(e_ is None for e_ in errs) else errs
                    if any(e_ is not None for e_ in errs):
                        why = "capture of the step with the all-reduce failed on rank(s) %s: %s" % (
                            [i for i, e_ in enumerate(errs) if e_ is not None], next(e_ for e_ in errs if e_ is not None))
                if why is None:
                    self._graph = graph
                    return
                graph = None
            warnings.warn("the gradient all-reduce stays OUT of the step graph on every rank (%s): eager exchange" % why)
            self.exchange_fallback = why
            self._opt_in_graph = False
            torch.cuda.synchronize()
        if not self.exchange:
            self._graph = self._capture_graph(self._opt_in_graph)
            return
        # N > 1: the last rung of the ladder.  A capture that was INVALIDATED (not merely refused) leaves torch's capture machinery
        # in a state in which no later capture on this process succeeds; if the plain forward + backward graph cannot be captured
        # on some rank, EVERY rank runs its steps eagerly (same kernels, same exchange, no graph) -- slower, never hung
        err, graph = None, None
        try:
            graph = self._capture_graph(False)
        except Exception as e:
            err = "%s: %s" % (type(e).__name__, str(e)[:160])
            self._abandon_capture(self._capturing)
        with self.watchdog.guard("agreement on the capture of the step graph", self.timeout_s):
            errs = self.ctl.gather(err)
        if any(e_ is not None for e_ in errs):
            why2 = "capture of forward + backward failed on rank(s) %s: %s" % ([i for i, e_ in enumerate(errs) if e_ is not None],
                                                                              next(e_ for e_ in errs if e_ is not None))
            warnings.warn("no step graph on any rank (%s): eager steps" % why2)
            self.exchange_fallback = (self.exchange_fallback + "; " if self.exchange_fallback else "") + why2
            self.exchange_checks["step_graph"] = errs
            self.use_graph = False
            self._graph = None
            return
        self._graph = graph

    def _abandon_capture(self, graph):
        """After a capture that raised: the graph object is never destroyed (_immortal), the capture's stream is taken out of
        capture mode if the failure left it there (an INVALIDATED capture: torch's capture_end throws before it ends the capture,
        and the thread stays in global capture mode -- every later eager launch fails), the device is drained."""
        _immortal(graph)
        cs = getattr(self, "_capture_stream", None)
        if cs is not None and self.bucket.flat.is_cuda:
            try:
                RF.H.stream_abandon_capture(cs)
            except Exception:
                pass
        try:
            torch.cuda.synchronize()
        except Exception:
            pass

    def _capture_mode(self):
        """hipStreamCaptureMode of the trainer's captures.  One rank: torch's default ("global": a forbidden call on ANY thread
        invalidates the capture).  With a process group: "thread_local" -- ProcessGroupNCCL's watchdog thread polls the events of
        earlier collectives (the parameter broadcast, the self-check's eager all-reduce) with hipEventQuery whenever it wakes up,
        and under "global" such a query from another thread, landing inside the capture window, invalidates a capture that did
        nothing wrong.  Kernels the autograd threads launch on the capturing stream are captured in either mode (capture is a
        property of the stream, the mode only decides whose forbidden calls count)."""
        return "thread_local" if self.exchange else "global"

    def _capture_graph(self, with_opt):
        graph = self._capturing = torch.cuda.CUDAGraph()
        # N > 1: every capture on a stream of its own (torch.cuda.graph otherwise re-uses ONE class-level capture stream: a capture
        # that failed on it would fail every later one)
        self._capture_stream = cs = RF.fresh_stream(self.bucket.flat.device) if self.exchange else None
        with torch.cuda.graph(graph, stream=cs, capture_error_mode=self._capture_mode()):
            self._loss = self._fwd_bwd(*self._static)
            if with_opt:
                if self.exchange:
                    self._graph_collective(self.bucket.flat)
                self._fused_opt.step_dev()              # (the 1/world factor is the hyper block's grad_scale)
        return graph

    def exchange_mode(self):
        """Where the gradient exchange of a step runs: 'none' (one rank), 'in-graph', 'eager'."""
        if not self.exchange:
            return "none"
        return "in-graph" if (self.use_graph and self._opt_in_graph) else "eager"

    def _check_first_graph_step(self):
        """After the FIRST replay of a step graph that holds the all-reduce: every rank must hold the same reduced gradient
        (checksum + norm over the control plane).  A captured collective that silently reduced nothing, or the wrong buffer, shows
        up here -- loudly -- instead of as replicas that drift apart."""
        self._first_graph_step_checked = True
        with self.watchdog.guard("first replay of the step graph with the in-graph all-reduce", self.timeout_s):
            torch.cuda.synchronize()
            f = self.bucket.flat
            sig = (float(f.double().sum().item()), float(f.double().norm().item()), float(self._fused_opt.norm.item()))
            # the comparison itself is on the BYTES of the reduced bucket (a 64-bit sum of its words: NaN-safe, order-independent
            # over the buffer, identical iff the ranks hold the same bits -- what a sum all-reduce of one buffer must leave)
            words = f.view(torch.int32).to(torch.int64)
            digest = int((words * (torch.arange(words.numel(), device=f.device, dtype=torch.int64) % 65521 + 1)).sum().item())
            finite = bool(torch.isfinite(f).all().item())
            sigs = self.ctl.gather((digest, finite, sig))
        self.exchange_checks["first_step_all_finite"] = all(s_[1] for s_ in sigs)
        self.exchange_checks["first_step_signatures_equal"] = all(s_[0] == sigs[0][0] for s_ in sigs)
        if not self.exchange_checks["first_step_all_finite"]:
            # (a NaN compares unequal to itself -- this used to be reported as "different gradients on the ranks")
            raise RuntimeError("non-finite gradient after the first replayed step on rank(s) %s (sum, norm, clip norm per rank: %r): "
                               "the model / batch diverged, not the exchange" % ([i for i, s_ in enumerate(sigs) if not s_[1]], [s_[2] for s_ in sigs]))
        if not self.exchange_checks["first_step_signatures_equal"]:
            raise RuntimeError("the in-graph gradient all-reduce left different BITS in the gradient bucket of the ranks (weighted word sum per rank: %r; "
                               "sum, norm, clip norm per rank: %r); re-run with RN_NO_GRAPH_ALLREDUCE=1" % ([s_[0] for s_ in sigs], [s_[2] for s_ in sigs]))

    def input_buffers(self, img, qst, label):
        """The captured step's OWN input tensors (captured now, from the given example batch, if it has not been yet).  A loader
        that writes every batch INTO these -- its host -> device copy, or a device-side producer -- and calls step(*buffers) hands
        the batch over with no device-to-device copy at all: step() copies only the tensors that are not these (one
        rn_copy_many launch, ~14 us for 12.6 MB of images in front of every replay).  Shapes are fixed by the example batch
        (a ragged last batch goes through step() with its own tensors and the eager path of the caller's choice).  The tensors
        stay the same objects across re-captures (the copy guard's switch to 16-bit copies).  Without a step graph there is
        nothing to hand over into: the arguments come back."""
        if not self.use_graph:
            return img, qst, label
        if self._graph is None:
            self._capture(img, qst, label)
        return self._static if self.use_graph else (img, qst, label)

    def step(self, img, qst, label):
        self._nstep += 1
        if self.copy_guard_every and ((self._nstep == 1 and self._guard_done_at is None) or self._nstep % self.copy_guard_every == 0):
            self.check_activation_copies(img, qst, label)
        if self.use_graph:
            if self._graph is None:
                self._capture(img, qst, label)
        if self.use_graph:                                  # (still: with N > 1 a capture that fails on any rank ends in eager steps everywhere)
            todo = [(dst, src) for dst, src in zip(self._static, (img, qst, label)) if dst.data_ptr() != src.data_ptr()]
            if todo and all(s_.is_cuda and s_.is_contiguous() and s_.dtype == d.dtype and s_.shape == d.shape
                                                     and s_.data_ptr() % 16 == 0 for d, s_ in todo):
                RF.H.copy_many(todo)                        # one launch for the whole batch hand-off
            else:
                for dst, src in todo:
                    dst.copy_(src, non_blocking=True)
            # BEFORE the replay, in every mode: the captured kernels have the parameters' addresses baked in
            if self._fused_opt is not None:
                self._fused_opt._check_storage()
            if self._opt_in_graph:
                self._fused_opt.sync_hyper(self.clip_norm, 1.0 / self.world)
            self._graph.replay()
            self.bucket.attach_()
            loss = self._loss
            if self._opt_in_graph:
                self._fused_opt.after_step_dev()
                if self.exchange and not self._first_graph_step_checked:
                    self._check_first_graph_step()
                return loss
        else:
            loss = self._fwd_bwd(img, qst, label)
        if self._fused_opt is not None:
            tm = self.timing                              # bench.py (N > 1): event brackets around the exchange and the optimiser
            if tm is not None:
                ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
                ev[0].record()
            gs = self.bucket.all_reduce_mean(self.group, scale=False, even_alone=self.exchange)
            if tm is not None:
                ev[1].record()
            self._fused_opt.step(self.clip_norm, gs)      # (1/world) + clip + Adam: two launches on the flat gradient
            if tm is not None:
                ev[2].record()
                tm.append(ev)
        else:
            self.bucket.all_reduce_mean(self.group, even_alone=self.exchange)
            if self.clip_norm:
                self.bucket.clip_grad_norm_(self.clip_norm)
            self.opt.step()
        return loss
