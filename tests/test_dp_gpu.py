"""GPU: the N > 1 code path of the data-parallel trainer on the hardware that is available (ONE MI355X).

  * FusedClipAdam.step(clip, grad_scale=1/world) on a world-times-summed bucket == the single-rank step, and both ==
    torch's clip_grad_norm_ + Adam (the `world > 1` branch of DataParallelTrainer.step, dp.py);
  * two processes on the same GPU, `gloo` carrying the CUDA bucket (RCCL refuses two ranks on one device): hipGraph
    replay of forward + backward, eager all-reduce, bucket re-attachment, fused (1/world + clip + Adam) -- the seam
    `bench.py --gpus N` / train.py run with N ranks -- against one process on the concatenated batch.
The 8-GPU RCCL run itself is the driver's (SCALE_rNN.json)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import formula

pytestmark = pytest.mark.gpu


class A:
    qdict_size, adict_size = formula.QDICT, formula.ADICT


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


# ---- FIRST when two devices are visible: the N > 1 path over RCCL, so that a failure there names its cause
# before anything else of this file runs.  (Helpers are defined further down: resolved at call time.)
@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs: the RCCL (backend nccl) all-reduce over xGMI")
@pytest.mark.parametrize("cfg", ["original-fp"])
def test_two_ranks_over_rccl_equal_one_rank_on_the_whole_batch(tmp_path, cfg):
    """The same assertion with one rank per DEVICE and backend "nccl" (= RCCL on ROCm) -- the configuration the reference's
    DataParallel wrap stands for (train.py:256-258) and the driver's SCALE run uses.  Runs wherever two GPUs are visible."""
    _check_two_ranks_against_one(tmp_path, cfg, "nccl")


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs: an asymmetric capture failure over RCCL")
def test_two_ranks_over_rccl_agree_on_the_fallback_when_one_capture_fails(tmp_path):
    """the same injected one-rank capture failure as test_exchange_fallback_is_agreed_by_all_ranks, but over RCCL on two
    devices -- the configuration in which a rank-local decision would hang the job inside a real ring kernel."""
    _check_two_ranks_against_one(tmp_path, "original-fp", "nccl", inject="rank1")


def test_backward_writes_gradients_into_the_bucket_only_when_autograd_assigns():
    """functional.grad_out: with a FlatGradBucket registered and .grad = None (the trainer's gather mode) every parameter
    gradient of the whole model is produced in its slot of the flat buffer -- gather_() launches nothing -- and equals, bitwise,
    what the fresh-tensor path gives; with .grad attached (accumulate mode) the slots are left alone and two backward passes
    add up."""
    import contextlib, io
    import relationnetworks_clevr_amd as pkg
    from relationnetworks_clevr_amd import dp
    from bench import make_batch
    import json
    hyp = json.load(open(os.path.join(os.path.dirname(pkg.rn_hip.__file__), "config.json")))["hyperparams"]["original-fp"]
    torch.manual_seed(3)
    with contextlib.redirect_stdout(io.StringIO()):
        model = pkg.RN(A, dict(hyp, dropout=0.0))
    model.cuda().train()
    img, qst, lab = make_batch(8, torch.device("cuda"), 128)
    bucket = dp.FlatGradBucket(model.parameters())

    def backward():
        torch.nn.functional.nll_loss(model(img, qst), lab).backward()

    bucket.detach_()
    backward()
    base = bucket.flat.data_ptr()
    for p_, o in zip(bucket.params, bucket.offsets):
        assert p_.grad is not None and p_.grad.data_ptr() == base + 4 * o, "a gradient was not produced in its slot"
    bucket.gather_()
    in_place = bucket.flat.clone()
    with pkg.options.override(grads_in_bucket=False):
        bucket.detach_()
        backward()
        assert all(p_.grad.data_ptr() != base + 4 * o for p_, o in zip(bucket.params, bucket.offsets))
        bucket.gather_()
    assert torch.equal(bucket.flat, in_place)
    # accumulate mode: .grad stays attached to the bucket, autograd adds into it -- the functions must NOT write there themselves
    bucket.zero_()
    backward()
    backward()
    torch.cuda.synchronize()
    err = (bucket.flat - 2 * in_place).abs().max() / in_place.abs().max()
    assert float(err) <= 1e-5, float(err)
    # TWO gradients for every parameter in ONE backward pass (the model run twice, the losses summed), gather mode: the slot may be
    # handed out once per pass -- the second backward function must get memory of its own, or autograd adds a buffer to itself
    img2, qst2, lab2 = make_batch(8, torch.device("cuda"), 128)
    img2 = img2.flip(0).contiguous()
    def grads_of(batches):
        bucket.detach_()
        sum(torch.nn.functional.nll_loss(model(i_, q_), l_) for i_, q_, l_ in batches).backward()
        bucket.gather_()
        torch.cuda.synchronize()
        return bucket.flat.clone()
    model.eval()                                          # (batch statistics of two passes would interact through the running buffers only)
    g1, g2 = grads_of([(img, qst, lab)]), grads_of([(img2, qst2, lab2)])
    both = grads_of([(img, qst, lab), (img2, qst2, lab2)])
    err = (both - (g1 + g2)).abs().max() / (g1 + g2).abs().max()
    assert float(err) <= 1e-5, float(err)


def test_fused_clip_adam_grad_scale_equals_single_rank_and_torch():
    import relationnetworks_clevr_amd as pkg
    from relationnetworks_clevr_amd import dp
    pkg.rn_hip.load()
    torch.manual_seed(1)

    def make():
        torch.manual_seed(5)
        return torch.nn.Sequential(torch.nn.Linear(300, 257), torch.nn.ReLU(), torch.nn.Linear(257, 31)).cuda()

    grads = None
    results = {}
    for tag, world in (("one", 1), ("two", 2), ("torch", 0)):
        m = make()
        opt = torch.optim.Adam(m.parameters(), lr=1e-3, weight_decay=1e-4)
        bucket = dp.FlatGradBucket(m.parameters())
        if grads is None:
            grads = [torch.randn(3, bucket.numel, device="cuda") * s for s in (0.01, 5.0, 40.0)]      # un-clipped and clipped steps
        if tag == "torch":
            for g in grads:
                for step_g in g:
                    bucket.flat.copy_(step_g)
                    torch.nn.utils.clip_grad_norm_(m.parameters(), 50.0)
                    opt.step()
        else:
            assert dp.FusedClipAdam.supports(bucket, opt)
            f = dp.FusedClipAdam(bucket, opt)
            for g in grads:
                for step_g in g:
                    bucket.flat.copy_(step_g * world)             # what a SUM all-reduce over `world` identical ranks leaves
                    f.step(50.0, grad_scale=1.0 / world)
        torch.cuda.synchronize()
        results[tag] = torch.cat([p.detach().reshape(-1) for p in m.parameters()]).cpu().numpy()
    assert np.allclose(results["one"], results["two"], rtol=0, atol=1e-7)       # x2 and x0.5 are exact in fp32
    assert np.allclose(results["one"], results["torch"], rtol=2e-5, atol=2e-7)


def _model(cfg, seed):
    import io, contextlib
    import relationnetworks_clevr_amd as pkg
    torch.manual_seed(seed)
    with contextlib.redirect_stdout(io.StringIO()):
        m = pkg.RN(A, dict(formula.HYP[cfg], dropout=0.0))
    m.cuda()
    m.train()
    if cfg.endswith("-fp"):
        m.conv.eval()          # batch statistics are per replica by design (train.py:256-258: no SyncBN): running stats here, so
    return m                   # that two shards of 4 ARE one batch of 8


# eps far above the gradient scale makes Adam's update ~ lr * g / eps, i.e. LINEAR in the gradient, and the clip is active:
# a wrong 1/world factor, a stale bucket or a clip on the un-scaled norm all show up in the parameters
CLIP = 0.5


def _adam(model):
    return torch.optim.Adam(model.parameters(), lr=1e-3, eps=1e-1, weight_decay=1e-4)


def _data(cfg, B):
    if cfg.endswith("-sd"):
        x = torch.from_numpy(formula.formula_objects(B, 12, 7, 5, from_pixels=False))
    else:
        x = torch.from_numpy(formula.hash_uniform((B, 3, 128, 128), 5, 0.0, 1.0))
    q = torch.from_numpy(formula.hash_ints((B, 12), 6, 1, formula.QDICT + 1))
    y = torch.from_numpy(formula.hash_ints((B,), 7, 0, formula.ADICT))
    return x.cuda(), q.cuda(), y.cuda()


def _trainer_class(inject):
    """inject = None: the product trainer.  "rank1": the in-graph exchange is ASKED for over gloo (graph_allreduce=True), the
    start-up self-check is replaced by a pass (gloo cannot be captured; the agreement call stays real) and the captured stand-in
    collective raises on rank 1 only -- rank 0's capture succeeds.  "selfcheck": the real self-check runs and fails (gloo)."""
    from relationnetworks_clevr_amd import dp
    if inject is None:
        return dp.DataParallelTrainer, {}

    class Injected(dp.DataParallelTrainer):
        def _graph_collective(self, tensor):
            if inject == "rank1":
                if self.rank == 1:
                    raise RuntimeError("injected capture failure on rank 1")
                return                                      # rank 0: a stand-in that captures fine
            super()._graph_collective(tensor)

        def _exchange_self_check(self):
            if inject == "rank1":
                self.ctl.gather(None)
                return None
            return super()._exchange_self_check()
    return Injected, {"graph_allreduce": True}


def _worker(rank, world, port, cfg, steps, out_path, backend="gloo", inject=None):
    from relationnetworks_clevr_amd import dp
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if backend == "nccl":                               # RCCL: one device per rank
        torch.cuda.set_device(rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    else:                                               # both ranks on device 0, gloo carries the bucket
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    model = _model(cfg, seed=3 + rank)                  # different init per rank: the broadcast must fix it
    opt = _adam(model)
    cls, kw = _trainer_class(inject)
    tr = cls(model, opt, clip_norm=CLIP, use_graph=True, timeout_s=240, **kw)
    assert tr._fused_opt is not None
    if inject is not None:
        assert tr._opt_in_graph and tr.exchange_mode() == "in-graph"      # (asked for; the first step decides)
    x, q, y = _data(cfg, 8)
    sh = x.shape[0] // world
    sl = slice(rank * sh, (rank + 1) * sh)
    losses = []
    for _ in range(steps):
        losses.append(float(tr.step(x[sl].contiguous(), q[sl].contiguous(), y[sl].contiguous()).detach()))
        tr.bucket.check_attached()
    lt = torch.tensor(losses, dtype=torch.float64)
    dist.all_reduce(lt)
    modes = tr.ctl.gather((tr.exchange_mode(), tr.exchange_fallback))
    if rank == 0:
        torch.save({"sd": {k: v.cpu() for k, v in model.state_dict().items()}, "loss": (lt / world).tolist(), "modes": modes,
                    "checks": tr.exchange_checks, "ranks_seen": tr.ctl.ranks_seen()}, out_path)
    else:
        tr.ctl.ranks_seen()
    dist.barrier()
    dist.destroy_process_group()


def _check_two_ranks_against_one(tmp_path, cfg, backend, inject=None):
    from relationnetworks_clevr_amd import dp
    steps = 3
    out = str(tmp_path / "dp.pt")
    try:
        mp.spawn(_worker, args=(2, _free_port(), cfg, steps, out, backend, inject), nprocs=2, join=True)
    except Exception as e:                                 # gloo built without device-tensor support: nothing to test here
        if backend == "gloo" and "gloo" in str(e).lower() and ("cuda" in str(e).lower() or "device" in str(e).lower()):
            pytest.skip("gloo cannot carry GPU tensors in this build: %s" % str(e)[:200])
        raise
    got = torch.load(out)
    assert got["ranks_seen"] == [0, 1]
    if inject is not None:
        # BOTH ranks ended up with the eager exchange and the same reason, although only one of them saw a failure
        assert [m_[0] for m_ in got["modes"]] == ["eager", "eager"], got["modes"]
        assert got["modes"][0][1] == got["modes"][1][1] and got["modes"][0][1]
        if inject == "rank1":
            assert "rank(s) [1]" in got["modes"][0][1] and "injected capture failure" in got["modes"][0][1]
            assert got["checks"]["capture"][0] is None and "injected" in got["checks"]["capture"][1]
        else:
            assert "self-check" in got["modes"][0][1]
    elif backend == "nccl":
        assert [m_[0] for m_ in got["modes"]] == ["in-graph", "in-graph"], got["modes"]
        assert got["checks"].get("self_check") == "passed" and got["checks"].get("first_step_signatures_equal") is True
    model = _model(cfg, seed=3)
    init = {k: v.detach().cpu().float().clone() for k, v in model.state_dict().items()}
    opt = _adam(model)
    tr = dp.DataParallelTrainer(model, opt, clip_norm=CLIP, use_graph=True)
    x, q, y = _data(cfg, 8)
    ref_loss = [float(tr.step(x, q, y).detach()) for _ in range(steps)]
    # The two shards compute what the whole batch computes question for question (no cross-question arithmetic), up to the
    # bf16-level noise of kernels that tile a 4-question and an 8-question problem differently, and the fp32 summation
    # order of the gradients.  Compared: the losses and every tensor's UPDATE (final - initial) in relative L2.
    assert np.allclose(got["loss"], ref_loss, rtol=2e-3, atol=1e-5), (got["loss"], ref_loss)
    errs = {}
    for k, v in model.state_dict().items():
        if not v.dtype.is_floating_point:
            continue
        da, db = got["sd"][k].float() - init[k], v.cpu().float() - init[k]
        if float(db.norm()) < 1e-9:                          # untouched tensor (conv stack of the *-sd model, BN buffers in eval)
            assert float(da.norm()) < 1e-9, k
            continue
        e = float((da - db).norm() / db.norm())
        errs[k] = (e, float(db.norm()), float(da.norm()))
    bad = {k: v for k, v in errs.items() if v[0] > 3e-2}
    assert not bad, "relative update error (err, |ref update|, |2-rank update|): %r" % (sorted(errs.items(), key=lambda kv: -kv[1][0])[:8],)


@pytest.mark.parametrize("cfg", ["original-sd", "original-fp"])
def test_two_ranks_on_one_gpu_equal_one_rank_on_the_whole_batch(tmp_path, cfg):
    _check_two_ranks_against_one(tmp_path, cfg, "gloo")


@pytest.mark.parametrize("inject", ["rank1"])
def test_exchange_fallback_is_agreed_by_all_ranks(tmp_path, inject):
    """VERDICT r4 item 1 / ADVICE r4: the in-graph exchange is asked for and the capture of the step fails on rank 1 ONLY while
    rank 0's succeeds.  The decision must be the job's: both ranks end up with the eager exchange, say why (naming rank 1), train on
    without hanging, and still equal one rank on the whole batch.  (Not a test: a REAL gloo all-reduce inside a capture --
    _trainer_class("selfcheck") -- invalidates the capture through a forbidden synchronisation; the self-check reports it and both
    ranks agree on the fallback, but HIP then refuses later launches of the process ("previous error during capture") even after
    the capture has been ended: the job fails LOUDLY with that error, it does not hang.  RCCL is built to be captured; gloo is never
    asked to in production (DataParallelTrainer only asks a stream-ordered backend).)"""
    _check_two_ranks_against_one(tmp_path, "original-fp", "gloo", inject=inject)


@pytest.mark.parametrize("cfg,B", [("original-fp", 64), ("ir-fp", 16), ("original-sd", 8)])
def test_captured_step_gradient_is_bitwise_reproducible(cfg, B):
    """The product step (BatchNorm in training mode, dropout on, the default arithmetic) replayed with lr = 0: the parameters stay
    put, so every replay must leave the SAME flat gradient bucket and the same loss, bit for bit -- no atomics, no launch-order
    dependence between the step's streams, no buffer shared across streams by accident (a race would show up as a replay that
    differs).  (The dropout mask is drawn per replay: dropout = 0 here so that the replays are the same function.)"""
    import contextlib, io, json
    import relationnetworks_clevr_amd as pkg
    from relationnetworks_clevr_amd import dp
    from bench import make_batch
    hyp = json.load(open(os.path.join(os.path.dirname(pkg.rn_hip.__file__), "config.json")))["hyperparams"][cfg]
    torch.manual_seed(5)
    with contextlib.redirect_stdout(io.StringIO()):
        model = pkg.RN(A, dict(hyp, dropout=0.0))
    model.cuda().train()
    img, qst, lab = make_batch(B, torch.device("cuda"), 128, state_desc=bool(hyp["state_description"]))
    opt = torch.optim.Adam(model.parameters(), lr=0.0, weight_decay=0.0)
    tr = dp.DataParallelTrainer(model, opt, clip_norm=50.0, use_graph=True, copy_guard_every=0)
    ref_flat = ref_loss = None
    names = [n_ for n_, p_ in model.named_parameters() if p_.requires_grad]
    for r in range(40):
        loss = tr.step(img, qst, lab)
        torch.cuda.synchronize()
        if ref_flat is None:
            ref_flat, ref_loss = tr.bucket.flat.clone(), float(loss.detach())
            assert float(ref_flat.abs().sum()) > 0
            continue
        if not torch.equal(tr.bucket.flat, ref_flat):
            bad = [n_ for n_, o, p_ in zip(names, tr.bucket.offsets, tr.bucket.params)
                   if not torch.equal(tr.bucket.flat[o:o + p_.numel()], ref_flat[o:o + p_.numel()])]
            raise AssertionError("replay %d: gradients differ from replay 0 in %r" % (r, bad))
        assert float(loss.detach()) == ref_loss


def _one_rank_rccl_worker(rank, port, cfg, steps, out_path, graph_allreduce):
    from relationnetworks_clevr_amd import dp
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    x, q, y = _data(cfg, 8)
    res = {}
    for name, kw in (("exchange", {"single_rank_exchange": True, "graph_allreduce": graph_allreduce, "timeout_s": 240}), ("plain", {})):
        model = _model(cfg, seed=3)
        # BatchNorm in TRAINING mode here (the product step: the fused conv + BN kernels, bitwise reproducible from replay to
        # replay -- tools/dbg/grad_repro.py).  _model()'s conv.eval() exists for the two-shard comparisons; it routes the conv
        # weight gradients through a path whose fp32 sums vary in their last bit from run to run (1e-7 relative)
        model.train()
        tr = dp.DataParallelTrainer(model, _adam(model), clip_norm=CLIP, use_graph=True, **kw)
        losses = [float(tr.step(x, q, y).detach()) for _ in range(steps)]
        tr.bucket.check_attached()
        torch.cuda.synchronize()
        res[name] = {"sd": {k: v.cpu() for k, v in model.state_dict().items()}, "loss": losses, "mode": tr.exchange_mode(),
                     "fallback": tr.exchange_fallback, "checks": tr.exchange_checks, "use_graph": tr.use_graph}
    torch.save(res, out_path)
    dist.destroy_process_group()


@pytest.mark.parametrize("graph_allreduce", [None, False], ids=["in-graph", "eager"])
def test_one_rank_rccl_exchange_runs_and_is_the_identity(tmp_path, graph_allreduce):
    """What a ONE-GPU box can execute of the N > 1 step over RCCL (VERDICT r4 "missing" 1: the in-graph all-reduce had never run
    anywhere): a one-rank `nccl` communicator, DataParallelTrainer(single_rank_exchange=True).  The trainer goes through exactly
    what it does with N > 1 -- start-up self-check (eager vs captured RCCL all-reduce of a bucket-sized buffer, bitwise), capture
    of forward + backward + all-reduce + clip / Adam into ONE hipGraph on a capture stream of its own, replay, first-replay
    signature check -- and, the sum over one rank being the identity, must land bit for bit on the plain one-rank trainer.
    (RCCL returns from an in-place all-reduce over ONE rank without launching a kernel: what runs here is torch's ProcessGroupNCCL
    under capture beside its watchdog thread, the capture stream, the agreement calls and the replay -- not the ring.)
    "eager": the same with the all-reduce launched behind the replayed forward + backward (the ladder's second rung; equal to
    1e-7: that rung's optimiser launch takes its scalars from the host)."""
    out = str(tmp_path / "one.pt")
    mp.spawn(_one_rank_rccl_worker, args=(_free_port(), "original-fp", 3, out, graph_allreduce), nprocs=1, join=True)
    got = torch.load(out)
    ex, pl = got["exchange"], got["plain"]
    assert pl["mode"] == "none"
    if graph_allreduce is None:
        assert ex["mode"] == "in-graph" and ex["fallback"] is None, (ex["mode"], ex["fallback"], ex["checks"])
        assert ex["checks"].get("self_check") == "passed" and ex["checks"].get("capture") == "ok"
        assert ex["checks"].get("first_step_signatures_equal") is True
    else:
        assert ex["mode"] == "eager" and ex["use_graph"], (ex["mode"], ex["fallback"])
    if graph_allreduce is None:
        assert ex["loss"] == pl["loss"], (ex["loss"], pl["loss"])
    else:       # the eager rung's optimiser takes its scalars (bias corrections, clip) from the host, the captured one from the device
        assert np.allclose(ex["loss"], pl["loss"], rtol=1e-6, atol=0), (ex["loss"], pl["loss"])
    for k, v in pl["sd"].items():
        if graph_allreduce is None or not v.dtype.is_floating_point:
            assert torch.equal(ex["sd"][k], v), k
        else:
            assert torch.allclose(ex["sd"][k], v, rtol=0, atol=1e-7), (k, float((ex["sd"][k] - v).abs().max()))


def test_bench_two_ranks_on_one_gpu():
    """The REAL N > 1 path of bench.py -- graph-replayed train step per rank, gradient all-reduce, fused 1/world + clip + Adam,
    barrier-bracketed timed region, MAX over ranks, kernel-timing passes on every rank, one JSON line from rank 0 -- launched
    exactly as the driver launches it (torch.distributed.run, 2 ranks), both ranks on this GPU with gloo carrying the bucket
    (RN_BENCH_SAME_GPU=1, RN_BENCH_BACKEND=gloo).  Throughput of two processes sharing one GPU is meaningless; the line's
    shape and bookkeeping are what is checked."""
    import json
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    env = dict(os.environ, RN_BENCH_SAME_GPU="1", RN_BENCH_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2", "--sustain", "1"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 5 and d["warmup"] == 2 and d["scaling"] == "weak"
    assert d["config"]["global_batch"] == 128 and d["config"]["parallelism"] == "dp2" and d["dtype"] == "f16s"
    assert abs(d["value"] - 2 * 64 * 5 / (d["ms_per_step"] * 5e-3)) < 1e-6 * d["value"]
    assert np.isfinite(d["loss"]) and "roofline" in d and "cpu_baseline" not in d and "parity" not in d
    assert d["sustained"]["steps"] >= 5 and d["sustained"]["value"] > 0            # (the same step count on both ranks: no hang)
    # a multi-rank line says where a step's time outside forward + backward goes (attribution of a scaling miss)
    assert d["allreduce_us_per_step"] > 0 and d["optimizer_us_per_step"] > 0 and d["allreduce_bytes"] == 4 * 484580
    # ... who took part, where the exchange ran and (had the in-graph form been asked for and refused) why not
    c = d["comm"]
    assert c["ranks_seen"] == [0, 1] and c["exchange_mode"] == "eager" and c["exchange_fallback"] is None and c["backend"] == "gloo"
    assert c["watchdog_s"] > 0


def test_bench_n_gt_1_path_over_one_rank_rccl():
    """bench.py's N > 1 code path -- process group over `nccl`, roll call, barriers, the trainer's in-graph exchange with its
    self-check, `comm` -- executed over RCCL on ONE device (RN_BENCH_ONE_RANK_EXCHANGE=1: a one-rank communicator; the sum over one
    rank is the identity).  What the driver's multi-GPU run adds to this is more ranks, nothing else in the script."""
    import json
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    env = dict(os.environ, RN_BENCH_ONE_RANK_EXCHANGE="1", MASTER_PORT=str(_free_port()))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--steps", "5", "--warmup", "2", "--sustain", "1", "--no-cpu-baseline",
           "--no-other-modes", "--no-parity"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["config"]["global_batch"] == 64 and np.isfinite(d["loss"])
    assert "all-reduce" in d["config"]["launch"]
    c = d["comm"]
    assert c["backend"] == "nccl" and c["ranks_seen"] == [0] and c["exchange_mode"] == "in-graph" and c["exchange_fallback"] is None
    assert c["exchange_checks"] == {"self_check": "passed", "capture": "ok", "first_step_all_finite": True, "first_step_signatures_equal": True}
    assert c["allreduce_bytes"] == 4 * 484580 and c["allreduce_us_per_step"] > 0 and c["watchdog_s"] > 0
    assert d["sustained"]["steps"] >= 5 and d["sustained"]["value"] > 0
