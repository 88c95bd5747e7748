"""The headline arithmetic over a TRAINING RUN, not four steps: `precision="auto"` (f16s forward, bf16 backward,
e4m3 activation copies) against fp32 on a learnable synthetic task, same seeds; and the guard that watches the e4m3 copies.
Reference: the training loop train.py:36-48 (mean NLL, clip 50, Adam with weight decay 1e-4)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    import relationnetworks_clevr_amd as p
    p.rn_hip.load()
    return p


def _first_below(curve, thr):
    for i, v in enumerate(curve):
        if v < thr:
            return i
    return len(curve)


def test_training_run_converges_like_fp32(pkg):
    """300 Adam steps (lr 1e-3, B = 64, graph-replayed DataParallelTrainer) of original-fp on train.SyntheticRelationalTask --
    one coloured square per image, the question asks for its colour or its quadrant -- in three arithmetic settings with the same
    seeds: fp32, the module default ("auto": f16s + e4m3 copies) and the default with 16-bit copies.  Stated band: every run
    starts at > 2 nats (28 answers: ln 28 = 3.3, the first 25-step window sits above it), reaches a 25-step mean loss < 0.05 and >= 95 %
    held-out accuracy; the default modes cross 0.1 within +-2 windows (50 steps) of fp32 and their area under the loss curve over
    the first 200 steps is within 35 % of fp32's.  (Later windows are not compared value by value: at this learning rate Adam
    runs show isolated loss spikes in ANY arithmetic -- the fp32 run has one around step 225.)"""
    from relationnetworks_clevr_amd import train as T
    runs = {"fp32": T.convergence_run("fp32", steps=300), "auto": T.convergence_run("auto", steps=300, h8=True),
            "auto16": T.convergence_run("auto", steps=300, h8=False)}
    for name, r in runs.items():
        c = r["loss"]
        assert c[0] > 2.0 and min(c) < 0.05 and r["accuracy"] >= 0.95, (name, c, r["accuracy"])
    ref = runs["fp32"]["loss"]
    t_ref, auc_ref = _first_below(ref, 0.1), float(np.sum(ref[:8]))
    for name in ("auto", "auto16"):
        c = runs[name]["loss"]
        assert abs(_first_below(c, 0.1) - t_ref) <= 2, (name, c, ref)
        assert abs(float(np.sum(c[:8])) - auc_ref) <= 0.35 * auc_ref, (name, c, ref)
    # the e4m3 guard looked at the copies at step 1 and left them alone: a default-initialised model flushes ~1 % of H_2
    g = runs["auto"]["copy_guard"]
    assert g and g[0]["step"] == 1 and not g[0]["switched"] and sorted(int(k) for k in g[0]["layers"]) == [0, 1, 2]
    assert all(v["flushed"] < 0.05 and v["clamped"] == 0.0 and 0 < v["max_value"] <= 448 for v in g[0]["layers"].values()), g[0]
    assert runs["auto16"]["copy_guard"] == [] and runs["fp32"]["copy_guard"] == []


def test_copy_health_counters_against_a_numpy_decode(pkg):
    """rn_fp8_copy_health on the outputs of a real forward chain: positive / flushed / clamped counts and the largest byte equal a
    numpy count over the decoded lane masks and the un-blocked e4m3 image (tests/test_gpu_kernels.py helpers)."""
    import test_gpu_kernels as K
    from oracle import formula
    H = pkg.rn_hip
    B, n, G, L, k, Q = 3, 64, 256, 4, 26, 128
    M, kt = B * n * n, 2 * 26 + 128
    x = formula.hash_uniform((B, n, k), 400, -1, 1).astype(np.float32)
    q = formula.hash_uniform((B, Q), 401, -1, 1).astype(np.float32)
    # (small weights: layer outputs of ~1e-3 .. 1e-1, so that some positive activations fall below e4m3's 2^-10)
    wd = [K.dev(formula.hash_uniform((G, kt if l == 0 else G), 410 + l, -0.02, 0.02).astype(np.float32)) for l in range(L)]
    bd = [K.dev(formula.hash_uniform((G,), 420 + l, -0.01, 0.01).astype(np.float32)) for l in range(L)]
    w0T = torch.empty(kt, G, device="cuda")
    hi, lo, jobs = K.f16s_images(H, wd, kt, k)
    H.pack_matrix_frag_many(jobs + [(wd[0], kt, 1, G, kt, w0T, 2)])
    Xp = torch.empty(B * n, 64, dtype=torch.float16, device="cuda"); Vc = torch.empty(B * n, G, device="cuda")
    H.pair_tables(K.dev(x), K.dev(q), w0T, bd[0], Xp, Vc, B, n, k, Q, G)
    Hs = [torch.empty(M, G, dtype=torch.float8_e4m3fn, device="cuda") for _ in range(L - 1)] + [None]
    masks = list(torch.zeros(L, H.g_chain_rr_mask_bytes(M), dtype=torch.uint8, device="cuda"))
    part = torch.empty(M // 256, G, device="cuda")
    H.g_chain_fwd_rr_f16s_alg0(Xp, Vc, n, hi, lo, bd, Hs, masks, part, M, G)
    torch.cuda.synchronize()
    for l in range(L - 1):
        got = H.fp8_copy_health(masks[l], Hs[l], M).cpu().tolist()
        gate = K.rr_mask_decode(masks[l], M, l)
        byt = K.unblock(Hs[l]).view(torch.uint8).cpu().numpy()
        want = [int(gate.sum()), int((gate & (byt == 0)).sum()), int((byt == 0x7e).sum()), int(byt.max())]
        assert got == want, (l, got, want)
        assert want[1] > 0                                   # (weights this small do flush something: the counter is exercised)


def test_copy_guard_switches_to_16_bit_copies(pkg, monkeypatch):
    """A model whose g-layer activations sit below e4m3's range at scale 1 (first g layer scaled down by 2^-12): the trainer's guard
    must see > 5 % of the positive activations flushed BEFORE the first capture, switch the module's own e4m3 switch off (not the
    process-wide option), and keep training -- into the same input tensors, with ONE capture."""
    from relationnetworks_clevr_amd import train as T, dp
    import contextlib, io, json, os
    hyp = dict(json.load(open(os.path.join(os.path.dirname(T.__file__), "config.json")))["hyperparams"]["original-fp"])

    class A:
        qdict_size, adict_size = 82, 28
    torch.manual_seed(3)
    with contextlib.redirect_stdout(io.StringIO()):
        m = pkg.RN(A, hyp)
    m.cuda(); m.train()
    with torch.no_grad():
        m.rl.g_layers[0].weight.mul_(2.0 ** -12); m.rl.g_layers[0].bias.mul_(2.0 ** -12)
    monkeypatch.setattr(pkg.options.OPT, "h8", True)
    tr = dp.DataParallelTrainer(m, torch.optim.Adam(m.parameters(), lr=1e-4), use_graph=True)
    img, qst, lab = T.load_tensor_data(next(iter(T.SyntheticRelationalTask(64, 64, seed=2))), "cuda")
    captures = []
    orig = tr._capture_graph
    monkeypatch.setattr(tr, "_capture_graph", lambda with_opt: (captures.append(with_opt), orig(with_opt))[1])
    with pytest.warns(UserWarning, match="16-bit copies"):
        bufs = tr.input_buffers(img, qst, lab)              # what train_epoch does first: the guard runs in front of the capture
    l0 = tr.step(*bufs).item()
    assert len(captures) == 1 and tr.input_buffers(img, qst, lab)[0] is bufs[0]
    assert pkg.options.OPT.h8 is True and m.rl._packed.h8 is False and tr.copy_guard_log[0]["switched"]
    assert tr.copy_guard_log[0]["layers"][0]["flushed"] > 0.05
    l1 = tr.step(*bufs).item()
    assert np.isfinite(l0) and np.isfinite(l1)
    # the switch is the module's: another model in the same process still gets e4m3 copies
    with contextlib.redirect_stdout(io.StringIO()):
        m2 = pkg.RN(A, hyp)
    assert m2.rl._packed.h8 is True


def test_many_trainers_in_one_process_keep_their_streams_apart(pkg):
    """Round 6: the 16th ir-fp trainer of one process died inside hipStreamEndCapture (stack overflow in hip::Stream::EndCapture).
    torch.cuda.Stream() hands out 32 pool streams round-robin, every model took one for its question encoder and every trainer one
    for its warm-up, so after 16 of them the encoder's stream WAS the weight-gradient stream it waits on by event -- a cyclic fork /
    join topology.  Role streams are now one per process and role (functional._side_stream) and never alias: 20 trainers, each
    captured and replayed, and the role streams are pairwise distinct hipStreams."""
    from relationnetworks_clevr_amd import train as T
    for i in range(20):
        r = T.convergence_run("auto", steps=3, batch=8, model_name="ir-fp", seed=i, eval_batches=1, log_every=1, task="pairs_dev")
        assert np.isfinite(r["loss"]).all(), (i, r["loss"])
    RF = pkg.functional
    handles = [s.cuda_stream for s in RF._SIDE_STREAMS.values()]
    assert len(handles) >= 4 and len(set(handles)) == len(handles), handles
    assert torch.cuda.current_stream().cuda_stream not in handles


def test_relational_task_plateau_exit_auto_vs_fp32_over_seeds(pkg):
    """does the default arithmetic leave the loss plateau of a RELATIONAL task when fp32 does?  ir-fp (question
    injected at layer 2: the model whose single-seed run of round 5 looked 12 points behind) on train.PairRelationTaskOnDevice
    (closest / same-row questions over three squares: the plateau at ~0.95 is "one-object questions solved, pair questions not"),
    six seeds x {fp32, auto}, 2200 Adam steps each (lr 5e-4, B = 64, clip 50, weight decay 1e-4: train.py:36-48).
    What the two 24-seed studies of the round measured (profiles/r06_convergence_seeds.txt, r06_convergence_seeds_run1.txt): exit
    step (trailing 250-step mean < 0.6) 1300 +- 281 for fp32 and 1350 +- 287 / 1464 +- 481 for auto, 2-3 of 24 runs of EITHER mode
    still on the plateau at 3000 steps; paired difference auto - fp32 pooled over both studies = +127 +- 54 (s.e.) steps -- a real,
    small lag (0.45 of the seed-to-seed s.d.) that the 16-bit-copies variant shares.  Stated window: at least 4 of the 6 runs of
    each mode have left the plateau by step 2200, and the MEDIAN exit steps (runs still on it count as 2200) differ by at most 600
    steps -- the measured lag plus more than two standard deviations of that difference for six seeds (per-seed s.d. 280 -> median
    143 -> difference 200)."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
    from convergence_seeds import exit_step
    from relationnetworks_clevr_amd import train as T
    steps, every = 2200, 50
    ex = {"fp32": [], "auto": []}
    for seed in range(1, 7):
        for mode in ex:
            r = T.convergence_run(mode, steps=steps, lr=5e-4, h8=True if mode == "auto" else None, task="pairs_dev", log_every=every,
                                  eval_batches=2, model_name="ir-fp", seed=seed)
            assert np.isfinite(r["loss"]).all() and r["loss"][0] > 1.2, (mode, seed, r["loss"][:3])
            ex[mode].append(exit_step(r["loss"], every))
    left = {m: sum(e is not None for e in v) for m, v in ex.items()}
    med = {m: float(np.median([e if e is not None else steps for e in v])) for m, v in ex.items()}
    assert left["fp32"] >= 4 and left["auto"] >= 4, ex
    assert abs(med["auto"] - med["fp32"]) <= 600, (med, ex)
