"""CPU, world_size 2 over gloo: the data-parallel host logic (flat gradient bucket, all-reduce
mean, clip, optimizer) gives the same parameters as one process on the concatenated batch
(SURVEY.md section 4 (iv) / 8e).  The model here is the oracle's CPU module -- the DP layer is
model-agnostic; the HIP path itself is exercised by the -m gpu tests."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import formula, rn_oracle as O


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _make(seed=3):
    torch.manual_seed(seed)
    hyp = dict(formula.HYP["original-sd"], dropout=0.0)
    return O.RNOracle(formula.QDICT, formula.ADICT, hyp)


def _data(B=4):
    x = torch.from_numpy(formula.formula_objects(B, 12, 7, 5, from_pixels=False))
    q = torch.from_numpy(formula.hash_ints((B, 9), 6, 1, formula.QDICT + 1))
    y = torch.from_numpy(formula.hash_ints((B,), 7, 0, formula.ADICT))
    return x, q, y


def _opt(model, kind):
    if kind == "sgd":      # (plain Adam turns 1e-9 round-off on zero-gradient weights into +-lr steps: its eps is raised below)
        return torch.optim.SGD(model.parameters(), lr=0.05, momentum=0.9, weight_decay=1e-4)
    # the reference's optimiser (train.py:330): Adam with coupled weight decay.  eps = 1e-4 keeps a weight whose true
    # gradient is 0 (summation-order noise of 1e-9) from moving by a full lr in a rank-dependent direction.
    return torch.optim.Adam(model.parameters(), lr=1e-3, eps=1e-4, weight_decay=1e-4)


def _worker(rank, world, port, steps, out_path, kind="sgd"):
    import relationnetworks_clevr_amd as pkg
    from relationnetworks_clevr_amd import dp
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    model = _make(seed=3 + rank)          # different init per rank: broadcast must fix it
    opt = _opt(model, kind)
    tr = dp.DataParallelTrainer(model, opt, clip_norm=50.0, use_graph=False)
    assert tr._fused_opt is None          # CPU tensors: the torch optimiser branch of DataParallelTrainer.step
    x, q, y = _data()
    sh = x.shape[0] // world
    sl = slice(rank * sh, (rank + 1) * sh)
    losses = []
    for _ in range(steps):
        losses.append(float(tr.step(x[sl], q[sl], y[sl]).detach()))
        tr.bucket.check_attached()
    lt = torch.tensor(losses, dtype=torch.float64)
    dist.all_reduce(lt)                    # mean of the shard losses == global-batch mean loss
    if rank == 0:
        torch.save({"sd": model.state_dict(), "loss": (lt / world).tolist()}, out_path)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("kind", ["sgd", "adam"])
def test_two_rank_dp_equals_single_process(tmp_path, kind):
    from relationnetworks_clevr_amd import dp
    steps = 3
    out = str(tmp_path / "dp.pt")
    mp.spawn(_worker, args=(2, _free_port(), steps, out, kind), nprocs=2, join=True)
    got = torch.load(out)
    # single process, full batch, same trainer (no process group -> all-reduce is the identity)
    model = _make(seed=3)
    opt = _opt(model, kind)
    tr = dp.DataParallelTrainer(model, opt, clip_norm=50.0, use_graph=False)
    x, q, y = _data()
    ref_loss = [float(tr.step(x, q, y).detach()) for _ in range(steps)]
    assert np.allclose(got["loss"], ref_loss, rtol=1e-5, atol=1e-6)
    for k, v in model.state_dict().items():
        assert torch.allclose(got["sd"][k], v, rtol=1e-4, atol=1e-6), k


def _ctl_worker(rank, world, port, out_dir):
    from relationnetworks_clevr_amd import dp
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ctl = dp.ControlPlane()
    res = {"all_true": ctl.all_ok(True), "one_false": ctl.all_ok(rank != 1), "any_one": ctl.any_of(rank == 1), "any_none": ctl.any_of(False),
           "gather": ctl.gather(None if rank == 0 else "boom on %d" % rank), "ranks_seen": ctl.ranks_seen()}
    torch.save(res, os.path.join(out_dir, "ctl%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_control_plane_decisions_are_the_jobs_not_the_ranks(tmp_path):
    """dp.ControlPlane: a condition that holds on ONE rank only -- a capture that throws, a self-check that
    mismatches -- must give the SAME answer on every rank, or the ranks end up in different exchange modes and hang."""
    mp.spawn(_ctl_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        res = torch.load(os.path.join(str(tmp_path), "ctl%d.pt" % r))
        assert res["all_true"] is True and res["one_false"] is False and res["any_one"] is True and res["any_none"] is False
        assert res["gather"] == [None, "boom on 1"] and res["ranks_seen"] == [0, 1]
    from relationnetworks_clevr_amd import dp
    one = dp.ControlPlane()                               # no process group: every decision is the local one
    assert one.world == 1 and one.all_ok(False) is False and one.all_ok(True) is True and one.ranks_seen() == [0] and one.gather("x") == ["x"]


def _one_rank_worker(rank, port, out_path):
    from relationnetworks_clevr_amd import dp
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=0, world_size=1)
    torch.set_num_threads(1)
    x, q, y = _data()
    res = {}
    for name, kw in (("exchange", {"single_rank_exchange": True}), ("plain", {})):
        model = _make(seed=3)
        tr = dp.DataParallelTrainer(model, _opt(model, "adam"), clip_norm=50.0, use_graph=False, **kw)
        losses = [float(tr.step(x, q, y).detach()) for _ in range(3)]
        res[name] = {"sd": {k: v.clone() for k, v in model.state_dict().items()}, "loss": losses, "mode": tr.exchange_mode(),
                     "exchange": tr.exchange, "timeout": tr.timeout_s}
    torch.save(res, out_path)
    dist.destroy_process_group()


def test_single_rank_exchange_is_the_identity(tmp_path):
    """DataParallelTrainer(single_rank_exchange=True) on a ONE-rank group runs the N > 1 exchange (here: the eager gloo all-reduce
    of the flat bucket, 1/world = 1) and lands bit for bit on the plain one-rank trainer -- the CPU half of
    tests/test_dp_gpu.py::test_one_rank_rccl_exchange_runs_and_is_the_identity.  Without a process group the flag is inert."""
    from relationnetworks_clevr_amd import dp
    out = str(tmp_path / "one.pt")
    mp.spawn(_one_rank_worker, args=(_free_port(), out), nprocs=1, join=True)
    got = torch.load(out)
    ex, pl = got["exchange"], got["plain"]
    assert ex["exchange"] is True and ex["mode"] == "eager" and ex["timeout"] > 0
    assert pl["exchange"] is False and pl["mode"] == "none" and pl["timeout"] == 0
    assert ex["loss"] == pl["loss"]
    for k, v in pl["sd"].items():
        assert torch.equal(ex["sd"][k], v), k
    m = _make()
    assert dp.DataParallelTrainer(m, _opt(m, "sgd"), single_rank_exchange=True).exchange is False     # (no process group here)


def test_watchdog_exits_instead_of_blocking():
    """dp.Watchdog: a wait that never ends becomes exit code 124 with a message naming what was waited for;
    a region that finishes in time is left alone."""
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    code = ("import sys, time; sys.path.insert(0, %r)\n"
            "from relationnetworks_clevr_amd import dp\n"
            "wd = dp.Watchdog(rank=3)\n"
            "dp.Watchdog.on_expire = lambda what, s: print('HOOK ' + what, flush=True)\n"
            "with wd.guard('quick region', 5.0):\n    time.sleep(0.05)\n"
            "print('survived', flush=True)\n"
            "with wd.guard('replay of the step graph', 0.3):\n    time.sleep(30)\n"
            "print('not reached', flush=True)\n") % root
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert p.returncode == 124, (p.returncode, p.stderr[-500:])
    assert "survived" in p.stdout and "not reached" not in p.stdout
    assert "HOOK replay of the step graph" in p.stdout and "HOOK quick" not in p.stdout      # (bench.py hangs its failure line on this hook)
    assert "rank 3" in p.stderr and "replay of the step graph" in p.stderr and "RN_NO_GRAPH_ALLREDUCE" in p.stderr


def test_flat_bucket_semantics():
    from relationnetworks_clevr_amd import dp
    m = torch.nn.Sequential(torch.nn.Linear(5, 3), torch.nn.Linear(3, 2))
    b = dp.FlatGradBucket(m.parameters())
    # every slot starts on 16 bytes (backward kernels write dW / db into the views with vector stores): 15 | 3 | 6 | 2 -> 16 + 4 + 8 + 2
    assert b.offsets == [0, 16, 20, 28] and b.numel == 30 == b.flat.numel()
    assert all(v.data_ptr() % 16 == 0 for v in b.views)
    x = torch.randn(7, 5)
    m(x).sum().backward()
    g1 = [p.grad.clone() for p in m.parameters()]
    b.check_attached()
    # clip == torch.nn.utils.clip_grad_norm_
    ref = [g.clone() for g in g1]
    tot = torch.sqrt(sum((g ** 2).sum() for g in ref))
    n = b.clip_grad_norm_(0.5)
    assert torch.allclose(n, tot)
    for p, g in zip(m.parameters(), ref):
        assert torch.allclose(p.grad, g * min(1.0, 0.5 / (float(tot) + 1e-6)), atol=1e-7)
    b.zero_()
    assert all(float(p.grad.abs().sum()) == 0 for p in m.parameters())
    # unused (slot zeroed, cached as zero) -> written IN PLACE by a backward kernel -> unused again: the slot must be
    # zeroed again, not trusted to be zero still
    w0, off0 = b.params[0], b.offsets[0]

    def others():                                          # parameters 1, 2: fresh tensors; parameter 3: written in its slot (so that
        b.detach_()                                        # gather_() takes the slot-by-slot branch, not the one concatenation)
        for p_ in b.params[1:3]:
            p_.grad = torch.ones_like(p_)
        p3, o3 = b.params[3], b.offsets[3]
        p3.grad = b.flat[o3:o3 + p3.numel()].view_as(p3)
    others()
    b.gather_()                                            # parameter 0 unused: zeroed once, remembered
    assert off0 in b._known_zero
    others()
    b.flat[off0:off0 + w0.numel()].fill_(7.0)              # what functional.grad_out's in-place writers do ...
    w0.grad = b.flat[off0:off0 + w0.numel()].view_as(w0)   # ... and hand autograd: a view of the slot itself
    b.gather_()
    assert off0 not in b._known_zero and float(w0.grad.sum()) == 7.0 * w0.numel()
    others()
    b.gather_()                                            # unused again: the stale 7s must go
    assert float(b.flat[off0:off0 + w0.numel()].abs().sum()) == 0.0
    b.zero_()
    m(x).sum().backward()                       # accumulates into the same views
    for p, g in zip(m.parameters(), g1):
        assert torch.allclose(p.grad, g)
    # gather mode: fresh gradients, one cat into the bucket, views re-attached
    b.detach_()
    assert all(p.grad is None for p in m.parameters())
    m(x).sum().backward()
    b.gather_()
    b.check_attached()
    for p, g in zip(m.parameters(), g1):
        assert torch.allclose(p.grad, g)
    torch.optim.SGD(m.parameters(), lr=0.1).zero_grad(set_to_none=True)
    with pytest.raises(RuntimeError, match="left the flat bucket"):
        b.check_attached()


def test_bench_launch_plumbing_dry_run_two_ranks():
    """bench.py under torch.distributed.run with 2 ranks, exactly as the driver launches the N > 1 bench, with the train
    step replaced by a sleep (RN_BENCH_DRY=1) and gloo instead of RCCL: WORLD_SIZE / RANK / MASTER_* parsing, process-group
    init, barrier-bracketed timed region, MAX-over-ranks time, one JSON line from rank 0 only."""
    import json
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    env = dict(os.environ, RN_BENCH_DRY="1", RN_BENCH_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["warmup"] == 1 and d["config"]["global_batch"] == 128
    assert d["ms_per_step"] >= 2.0          # rank 1 sleeps 2 ms per step, rank 0 only 1 ms: the MAX over ranks won
    assert abs(d["value"] - 2 * 64 * 4 / (d["ms_per_step"] * 4e-3)) < 1e-6 * d["value"]


def test_bench_prints_one_failure_line_whatever_goes_wrong():
    """the first real N > 1 run is a one-shot, so a bench that fails must still leave ONE JSON line on rank 0 that
    says how far it got (`failed`, `error`, `phase`, `n_gpus`, `comm.exchange_mode / exchange_fallback / ranks_seen /
    allreduce_us_per_step`).  (a) a launch without torch.distributed.run; (b) two ranks over gloo (dry run) of which rank 1 raises:
    torch.distributed.run tears rank 0 down with SIGTERM, and rank 0's handler prints the line."""
    import json
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=300,
                       env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")})
    assert p.returncode != 0
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["failed"] is True and d["value"] is None and "torch.distributed.run" in d["error"]
    assert set(d["comm"]) >= {"exchange_mode", "exchange_fallback", "ranks_seen", "allreduce_us_per_step"}
    env = dict(os.environ, RN_BENCH_DRY="1", RN_BENCH_BACKEND="gloo", RN_BENCH_DRY_FAIL_RANK="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "400", "--warmup", "1"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, (p.stdout, p.stderr[-1500:])
    d = json.loads(lines[0])
    # (rank 0 either sees its collective break -- "Connection closed by peer" -- or is torn down by the launcher's SIGTERM first)
    assert d["failed"] is True and d["n_gpus"] == 2 and d["steps"] == 400 and d["error"] and "dry run" in d["phase"]
