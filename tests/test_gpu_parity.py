"""GPU: the drop-in module (RelationalLayer / RN on HIP kernels) against the golden vectors
recorded from the reference (tests/golden/*.npz).

Tolerances (every bound is <= ~2x what was measured on MI355X; measured values are appended to
gpurun_out/parity_report.jsonl by every run):
  precision="fp32" (fp32 MFMA): log-probs <= 1e-4 max-norm relative (the north star asks for 1e-3; measured
      <= 5e-7); parameter grads <= 1e-3; input grads <= 2e-4 in relative L2 and <= 5e-3 max-norm: among the
      10^7..10^8 g_theta ReLU units of a batch a handful sit within fp32 round-off of zero and gate differently
      under a different (equally valid) fp32 summation order; one flipped unit moves one object's dx by ~1e-3 of
      max|dx| (measured: sparse (b, j) rows at 6e-4..1.3e-3, everything else at 1e-6);
  precision="f16s" / "auto" (the HEADLINE mode: fp16 activations x split fp16 weights, bf16 backward): log-probs
      <= 1e-3 = the contract (CONTRACT) and, separately, <= 3e-4 = a regression guard at 1.5x the worst measured value
      (F16S_GUARD; measured 2e-6..2.0e-4), argmax agreement with the reference = 1.0, gradients
      <= 1.2e-2 in relative L2 (measured 2e-3..5e-3: bf16 storage of dZ / H in the backward pass);
  precision="bf16" (single-pass bf16, the throughput mode -- NOT the headline): log-probs <= 3e-3 with formula
      weights (measured 0.4e-3..1.5e-3) and <= 2e-2 on the released checkpoints (measured 0.9e-2..1e-2: the
      systematic weight-rounding error SURVEY.md appendix B predicted; this mode misses the 1e-3 bar and says so);
      gradients per fixture in BF16_GRAD_L2 (a bf16-sized perturbation of x_g flips f_phi / g_theta ReLU units that
      sit near zero, which switches whole gradient contributions discontinuously -- worst on the small-batch
      fixtures that run the per-layer kernels).  The per-kernel bf16 tests in test_gpu_kernels.py are the tight ones
      (<= 1 bf16 ulp against an oracle on the same operands)."""
import json
import os

import numpy as np
import pytest
import torch

import gold
from oracle import formula

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
# Two bounds on the headline mode's log-probs, on purpose (a bound set AT the measurement is a coin flip):
CONTRACT = 1e-3       # north_star: "logits matching the reference PyTorch path within 1e-3 relative fp32 tolerance" -- the parity bar
# ... and a REGRESSION GUARD that says the arithmetic has not got worse: 1.5x the worst value measured on MI355X over rounds 3-5
# (1.99e-4: the released ir-fp checkpoint in the training arithmetic; G-fp64 1.6e-4; profiles/r05_parity_report.jsonl has every value).
# A failure of the guard alone means "look at what changed", not "parity lost".
F16S_GUARD = 3e-4
TWO_PASS_GUARD = 1.5e-4     # the eval() arithmetic (hi + lo on every layer): measured <= 7e-5
RL_TAGS = ["G-sd4", "G-irsd4", "G-fp-small", "G-ir-small", "G-fp64", "G-ir64", "G-fp196", "G-drop"]
# relative-L2 gradient bounds of the bf16 mode, ~2x the measured value per fixture (dx, dq, bias grads)
BF16_GRAD_L2 = {"G-drop": 3e-2, "G-fp-small": 3e-2, "G-fp196": 3e-2, "G-fp64": 8e-2, "G-sd4": 8e-2, "G-irsd4": 0.12,
                "G-ir64": 8e-2, "G-ir-small": 0.2}     # (B = 2 on the per-layer bf16 kernels: one flipped f_phi unit is half the batch's gradient; fp32 on the same path: 2e-4)


def report(tag, **kw):
    d = os.path.join(ROOT, "gpurun_out")
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "parity_report.jsonl"), "a") as f:
        f.write(json.dumps(dict(tag=tag, **kw)) + "\n")
    print("PARITY", tag, kw)


@pytest.fixture(scope="module")
def pkg():
    import relationnetworks_clevr_amd as p
    p.rn_hip.load()
    torch.cuda.set_device(0)
    return p


def l2rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def run_rl(pkg, g, precision):
    meta = g["meta"]
    hyp, sd, x, q, lab = gold.rl_case(meta)
    hyp = dict(hyp, precision=precision)
    rl = pkg.RelationalLayer(hyp["rl_in_size"], formula.ADICT, hyp["lstm_hidden"], hyp)
    rl.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    rl = rl.cuda()
    if "dropout_mask" in g:
        rl.train()
        rl.forced_dropout_mask = torch.from_numpy(g["dropout_mask"]).cuda()
    else:
        rl.eval()
    if meta.get("strided"):
        xt = torch.from_numpy(np.ascontiguousarray(x.transpose(0, 2, 1))).cuda().permute(0, 2, 1)
    else:
        xt = torch.from_numpy(x).cuda()
    xt.requires_grad_(True)
    qt = torch.from_numpy(q).cuda().requires_grad_(True)
    lp = rl(xt, qt)
    loss = torch.nn.functional.nll_loss(lp, torch.from_numpy(lab).cuda())
    loss.backward()
    torch.cuda.synchronize()
    grads = {n: p.grad.detach().cpu().numpy() for n, p in rl.named_parameters()}
    return lp.detach().cpu().numpy(), float(loss.detach()), xt.grad.cpu().numpy(), qt.grad.cpu().numpy(), grads


@pytest.mark.parametrize("tag", RL_TAGS)
def test_relational_layer_fp32_parity(pkg, tag):
    g = gold.load(tag)
    lp, loss, dx, dq, grads = run_rl(pkg, g, "fp32")
    e_lp, e_dx, e_dq = gold.rel_err(lp, g["log_probs"]), gold.rel_err(dx, g["dx"]), gold.rel_err(dq, g["dq"])
    rep = {}
    e_w = gold.check_grads(g, grads, 1e-3, rep)
    report(tag, precision="fp32", log_probs=e_lp, dx=e_dx, dq=e_dq, params=e_w)
    assert e_lp <= 1e-4
    assert abs(loss - float(g["loss"])) <= 1e-4 * max(1.0, abs(float(g["loss"])))
    assert e_dx <= 5e-3 and e_dq <= 5e-3
    assert l2rel(dx, g["dx"]) <= 2e-4 and l2rel(dq, g["dq"]) <= 2e-4


@pytest.mark.parametrize("tag", RL_TAGS)
def test_relational_layer_bf16_parity(pkg, tag):
    g = gold.load(tag)
    lp, loss, dx, dq, grads = run_rl(pkg, g, "bf16")
    e_lp = gold.rel_err(lp, g["log_probs"])
    e_dx, e_dq = l2rel(dx, g["dx"]), l2rel(dq, g["dq"])
    e_b = max(l2rel(grads[k[5:]], g[k]) for k in g if k.startswith("grad/"))
    report(tag, precision="bf16", log_probs=e_lp, dx_l2=e_dx, dq_l2=e_dq, params_l2=e_b,
           dx_max=gold.rel_err(dx, g["dx"]), argmax_agree=float((lp.argmax(1) == g["log_probs"].argmax(1)).mean()))
    assert np.isfinite(lp).all()
    assert e_lp <= 3e-3
    bound = BF16_GRAD_L2[tag]
    assert e_dx <= bound and e_dq <= bound and e_b <= bound, (e_dx, e_dq, e_b, bound)


@pytest.mark.parametrize("tag", RL_TAGS)
def test_relational_layer_bf16x3_parity(pkg, tag):
    """precision="bf16x3": fp32 storage, the g_theta forward / dgrad products as three bf16 MFMA products of operands split into
    hi + lo while they are staged (rn_gemm.hip, RN_F32X3) -- the 16-bit arithmetic of the 512-wide *-sd models, whose layers the
    register-resident chains do not cover (the weight gradients run the same arithmetic, rn_wgrad.hip; f_phi stays exact fp32).  Contract: 1e-3 on log-probs; what it
    measures is ~1e-5 (the bounds here are regression guards an order of magnitude above the measured values)."""
    g = gold.load(tag)
    lp, loss, dx, dq, grads = run_rl(pkg, g, "bf16x3")
    e_lp, e_dx, e_dq = gold.rel_err(lp, g["log_probs"]), l2rel(dx, g["dx"]), l2rel(dq, g["dq"])
    e_b = max(l2rel(grads[k[5:]], g[k]) for k in g if k.startswith("grad/"))
    report(tag, precision="bf16x3", log_probs=e_lp, dx_l2=e_dx, dq_l2=e_dq, params_l2=e_b, dx_max=gold.rel_err(dx, g["dx"]))
    assert e_lp <= 1e-3                                   # the contract
    assert e_lp <= 1e-4 and e_dx <= 1e-3 and e_dq <= 1e-3 and e_b <= 1e-3, (e_lp, e_dx, e_dq, e_b)   # regression guards


@pytest.mark.parametrize("tag", ["G-fp-small", "G-fp64", "G-drop", "G-ir-small", "G-ir64"])
def test_relational_layer_f16s_parity(pkg, tag):
    """precision="f16s" (fp16 activations x split fp16 weights, bf16 backward): the FAST mode that meets
    the north-star bar -- log-probs <= 1e-3 max-norm relative (measured 7e-5..2.5e-4); gradients are
    bf16-class (same metric and bound as the bf16 mode)."""
    g = gold.load(tag)
    lp, loss, dx, dq, grads = run_rl(pkg, g, "f16s")
    e_lp = gold.rel_err(lp, g["log_probs"])
    e_dx, e_dq = l2rel(dx, g["dx"]), l2rel(dq, g["dq"])
    e_b = max(l2rel(grads[k[5:]], g[k]) for k in g if k.startswith("grad/"))
    agree = float((lp.argmax(1) == g["log_probs"].argmax(1)).mean())
    # every parameter gradient the fixture pins -- full tensors (grad/), or 64 sampled entries + the norm (gradsample/, gradnorm/:
    # the full-size fixtures' weight gradients, i.e. what the e4m3 activation copies touch) -- in the max-norm metric of gold.py
    per = {}
    e_w = gold.check_grads(g, grads, 3e-2, per)
    report(tag, precision="f16s", log_probs=e_lp, dx_l2=e_dx, dq_l2=e_dq, params_l2=e_b, argmax_agree=agree, param_grads_max=e_w,
           g_weight_grads={k: v for k, v in per.items() if k.startswith("g_layers") and k.endswith("weight")})
    assert e_lp <= CONTRACT
    assert e_lp <= F16S_GUARD, "regression guard (the contract, 1e-3, still holds)"
    assert agree == 1.0
    # gradients: the backward pass of a ReLU network depends on the forward pass through the GATES only; one-pass fp16 weights
    # (2^-12 relative per row -- the dithering averages over tiles, not inside a row) flip the gate of ~1e-3 of the units, those
    # whose pre-activation is rounding noise.  Measured dx / dq 1.4e-2 / 1.2e-2 on G-fp64 (two passes on every layer: 4.7e-3 /
    # 2.9e-3; the bf16 mode: up to 1.2e-1, BF16_GRAD_L2) -- noise, not bias: test_training_trajectory pins the consequence.
    assert e_dx <= 2e-2 and e_dq <= 2e-2 and e_b <= 1.2e-2, (e_dx, e_dq, e_b)
    if g["dx"].shape[0] >= 32:
        # per QUESTION (tools/dbg/grad_error_split.py): the whole-batch figures above are carried by the one or two
        # questions in which an f_phi ReLU flips (8 - 13 % on G-fp64: one unit of a 256-wide layer); the median is what the arithmetic
        # does -- flipped g_theta gates ~9e-3 on dx, the bf16 backward chain ~3e-3 on both
        pq = lambda a, r: np.array([l2rel(a[i], r[i]) for i in range(a.shape[0])])
        mx, mq = float(np.median(pq(dx, g["dx"]))), float(np.median(pq(dq, g["dq"])))
        report(tag, precision="f16s", dx_l2_median_per_question=mx, dq_l2_median_per_question=mq)
        assert mx <= 1.5e-2 and mq <= 6e-3, (mx, mq)


@pytest.mark.parametrize("tag", ["G-fp64", "G-fp-small", "G-drop", "G-ir64", "G-ir-small", "G-sd4", "G-irsd4", "G-fp196"])
def test_headline_mode_is_the_module_default_and_meets_the_bar(pkg, tag):
    """precision="auto" -- what a user who touches nothing gets, and what bench.py reports as `value` -- is parity-clean on EVERY
    fixture: "f16s" on the headline shape family (four 256-wide g layers -- the 14 x 14 grid, n = 196, on the padded j axis),
    "fp32" where no f16s kernel covers the shape (the 512-wide *-sd models of config.json) -- never single-pass bf16.
    Log-probs within the contract (1e-3) and the regression guard (3e-4) of the reference, same answers."""
    g = gold.load(tag)
    hyp = formula.HYP[g["meta"]["cfg"]]
    rl = pkg.RelationalLayer(hyp["rl_in_size"], formula.ADICT, hyp["lstm_hidden"], dict(hyp))
    resolved = rl.resolved_precision(g["meta"]["b"], g["meta"]["n"], hyp["rl_in_size"] // 2)
    assert rl.precision == "auto" and resolved == ("fp32" if tag in ("G-sd4", "G-irsd4") else "f16s")
    lp, loss, dx, dq, grads = run_rl(pkg, g, "auto")
    e_lp = gold.rel_err(lp, g["log_probs"])
    report(tag, precision="auto", resolved=resolved, log_probs=e_lp)
    assert e_lp <= CONTRACT and (lp.argmax(1) == g["log_probs"].argmax(1)).all()
    assert e_lp <= F16S_GUARD, "regression guard (the contract, 1e-3, still holds)"


@pytest.mark.parametrize("tag,precision", [("G-fp64", "f16s"), ("G-ir64", "f16s")])
def test_e4m3_activation_copies_touch_only_the_g_weight_gradients(pkg, tag, precision, monkeypatch):
    """The factored-first-layer chains keep H_0..2 for the weight gradient as e4m3 bytes (RN_H8=0: 16-bit copies).  Nothing but
    dW of g layers 1..3 reads them: log-probs, dx, dq, the bias gradient of layer 0, the f_phi gradients and dW_0 (pair
    reductions) must be bitwise those of the 16-bit copies, the bias gradients of layers 1, 2 the same sums in another fp32 order
    (<= 2e-6: the wide units' in-lane column sums), and the three weight gradients within 3e-3 (relative L2).  The LAST
    layer's bias gradient moves too (<= 3e-3): with e4m3 copies its gradient matrix is never formed -- the gate job of
    rn_g_wgrad_blocked scales the gate sums by the un-rounded dxg -- while the 16-bit path stores bf16(dxg) x gate.  The error of
    every touched tensor against the fp32 reference (sampled entries + norm for the full-size fixtures) is measured and reported,
    and must stay inside the mode's own band."""
    g = gold.load(tag)
    # (both legs with a stored dZ_0: the chain that reduces it on chip needs the e4m3 set -- its own test is the next one)
    monkeypatch.setattr(pkg.options.OPT, "chain_reduce", False)
    monkeypatch.setattr(pkg.options.OPT, "h8", False)
    lp0, loss0, dx0, dq0, gr0 = run_rl(pkg, g, precision)
    monkeypatch.setattr(pkg.options.OPT, "h8", True)
    lp1, loss1, dx1, dq1, gr1 = run_rl(pkg, g, precision)
    assert np.array_equal(lp0, lp1) and np.array_equal(dx0, dx1) and np.array_equal(dq0, dq1)
    touched = {"g_layers.%d.weight" % l for l in (1, 2, 3)} | {"g_layers.3.bias"}
    # round 6: on e4m3 images the stored-gradient jobs run as WIDE units, whose db is an in-lane sum of the same bf16 dZ values
    # (v_dot2c_f32_bf16) over other row splits than the quad units' MFMA against ones: the bias gradients of layers 1, 2 are the
    # same sums in another fp32 order
    reordered = {"g_layers.1.bias", "g_layers.2.bias"}
    rep = {}

    def ref_err(name, arr):
        """error against the reference: relative L2 where the fixture keeps the tensor, else worst of (sampled entries, norm)"""
        if "grad/" + name in g:
            return l2rel(arr, g["grad/" + name])
        sub = {k_: v for k_, v in g.items() if k_.endswith("/" + name)}
        out = {}
        gold.check_grads(sub, {name: arr}, 1.0, out)
        return out[name]
    for k in gr0:
        if k in touched:
            d = l2rel(gr1[k], gr0[k])
            rep[k] = (d, ref_err(k, gr0[k]), ref_err(k, gr1[k]))
            assert 0 < d <= 3e-3, (k, d)
            assert rep[k][2] <= max(2.0 * rep[k][1], 4e-3), (k, rep[k])     # e4m3 copies stay in the 16-bit copies' own error class
        elif k in reordered:
            assert l2rel(gr1[k], gr0[k]) <= 2e-6, (k, l2rel(gr1[k], gr0[k]))
        else:
            assert np.array_equal(gr0[k], gr1[k]), k
    report(tag, precision=precision, e4m3_vs_16bit={k: v[0] for k, v in rep.items()}, ref_err_16bit={k: v[1] for k, v in rep.items()},
           ref_err_e4m3={k: v[2] for k, v in rep.items()})


@pytest.mark.parametrize("tag", ["G-fp64", "G-ir64", "G-fp-small", "G-fp196"])
def test_pair_reductions_inside_the_backward_chain(pkg, tag, monkeypatch):
    """rn_g_chain_bwd_rr_red (the module default on the chain path, the padded j axis of the 14 x 14 grid included -- G-fp196):
    layer 0's gradient never leaves the chip, its pair-axis sums are
    formed in fp32 from the un-rounded accumulators.  Against the stored-dZ_0 path (bf16 rows + rn_pair_reduce_bwd,
    RN_NO_CHAIN_REDUCE=1): the forward and everything that does not read those sums -- log-probs, dW / db of layers 1..3, f_phi --
    is bitwise the same; dx, dq (question at layer 0), dW_0, db_0 move by the bf16 rounding that is gone (<= 5e-3 relative L2) and
    do not get worse against the reference (model.py:117-127 backward)."""
    g = gold.load(tag)
    monkeypatch.setattr(pkg.options.OPT, "chain_reduce", False)
    lp0, loss0, dx0, dq0, gr0 = run_rl(pkg, g, "f16s")
    monkeypatch.setattr(pkg.options.OPT, "chain_reduce", True)
    lp1, loss1, dx1, dq1, gr1 = run_rl(pkg, g, "f16s")
    assert np.array_equal(lp0, lp1)
    inj0 = formula.HYP[g["meta"]["cfg"]]["question_injection_position"] == 0
    assert 0 < l2rel(dx1, dx0) <= 5e-3
    assert (0 < l2rel(dq1, dq0) <= 5e-3) if inj0 else np.array_equal(dq0, dq1)
    for k in gr0:
        if k.startswith("g_layers.0."):
            assert 0 < l2rel(gr1[k], gr0[k]) <= 5e-3, k
        else:
            assert np.array_equal(gr0[k], gr1[k]), k
    e0, e1 = l2rel(dx0, g["dx"]), l2rel(dx1, g["dx"])
    report(tag, precision="f16s", chain_reduce_dx_l2=(e0, e1), chain_reduce_dq_l2=(l2rel(dq0, g["dq"]), l2rel(dq1, g["dq"])))
    assert e1 <= 1.05 * e0 + 1e-4


def test_injected_layer_question_sums_from_the_wgrad_partials(pkg, monkeypatch):
    """ir-*: the per-question sums of the injected layer's gradient (Rq -> dq and the question columns of dW_2) come from the
    streaming wgrad kernel's per-split column sums when its row splits can be question-aligned, else from a pass over dZ_2
    (rn_blocked_question_sums; forced here by hiding the aligned split count): the same bf16 values added in fp32 in another
    order -- everything agrees to fp32 rounding; what the blocked weight-gradient launch does not produce is bitwise the same (its
    row splits are question-aligned only when Rq is taken from them: other summation order for dW / db of layers 1..3)."""
    g = gold.load("G-ir64")
    real = pkg.rn_hip.wgrad_blocked_splits
    monkeypatch.setattr(pkg.rn_hip, "wgrad_blocked_splits", lambda M, rpq=0, njobs=1, aligned=False: 0 if aligned else real(M, rpq, njobs, aligned))
    lp0, loss0, dx0, dq0, gr0 = run_rl(pkg, g, "f16s")
    monkeypatch.setattr(pkg.rn_hip, "wgrad_blocked_splits", real)
    lp1, loss1, dx1, dq1, gr1 = run_rl(pkg, g, "f16s")
    assert np.array_equal(lp0, lp1) and np.array_equal(dx0, dx1)
    assert 0 < l2rel(dq1, dq0) <= 1e-5
    for k in gr0:
        if k.startswith("g_layers.") and not k.startswith("g_layers.0."):
            assert l2rel(gr1[k], gr0[k]) <= 1e-5, k
            if k == "g_layers.2.weight":
                assert 0 < l2rel(gr1[k][:, 256:], gr0[k][:, 256:]) <= 1e-5
        else:
            assert np.array_equal(gr0[k], gr1[k]), k


def test_f16s_refuses_unsupported_shapes(pkg):
    g = gold.load("G-irsd4")                      # 512-wide g layers: no fused chain
    with pytest.raises(RuntimeError, match="f16s"):
        run_rl(pkg, g, "f16s")


def build_full(pkg, g, precision, **extra):
    meta = g["meta"]
    hyp = dict(formula.HYP[meta["cfg"]], precision=precision, **extra)

    class Args:
        qdict_size = formula.QDICT
        adict_size = formula.ADICT

    m = pkg.RN(Args, hyp)
    return m, meta


@pytest.mark.parametrize("tag", ["G-e2e", "G-e2e-ir"])
@pytest.mark.parametrize("precision,tol", [("fp32", 1e-5), ("bf16", 3e-3), ("f16s", F16S_GUARD), ("f16s-eval", TWO_PASS_GUARD)])
def test_full_model_e2e(pkg, tag, precision, tol):
    """"f16s": the TRAINING arithmetic (eval_two_pass off -- what bench.py times); "f16s-eval": what eval() runs by default."""
    g = gold.load(tag)
    m, meta = build_full(pkg, g, precision.split("-")[0], eval_two_pass=precision.endswith("-eval"))
    shapes = {k: tuple(v) for k, v in json.loads(str(g["state_names"])).items()}
    sd = formula.formula_fill_state(shapes, meta["seed"])
    res = m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    assert not res.unexpected_keys and all(k.endswith("num_batches_tracked") for k in res.missing_keys)
    m.cuda(); m.eval()
    img = torch.from_numpy(formula.hash_uniform((meta["b"], 3, meta["img_hw"], meta["img_hw"]), meta["seed"] + 1, 0.0, 1.0)).cuda()
    qst = torch.from_numpy(formula.hash_ints((meta["b"], meta["T"]), meta["seed"] + 2, 1, formula.QDICT + 1)).cuda()
    with torch.no_grad():
        lp = m(img, qst).cpu().numpy()
        conv = m.conv(img).cpu().numpy()
        qe = m.text(qst).cpu().numpy()
    e = gold.rel_err(lp, g["log_probs"])
    report(tag, precision=precision, log_probs=e, conv=gold.rel_err(conv, g["conv_out"]), qst=gold.rel_err(qe, g["qst_emb"]))
    assert e <= (CONTRACT if precision.startswith("f16s") else tol)
    assert e <= tol, "regression guard"


@pytest.mark.parametrize("tag,precision,tol", [("pretrained_original_fp", "fp32", 2e-6), ("pretrained_ir_fp", "fp32", 2e-6),
                                               ("pretrained_original_fp", "bf16", 2e-2), ("pretrained_ir_fp", "bf16", 2e-2),
                                               ("pretrained_original_fp", "f16s", F16S_GUARD), ("pretrained_original_fp", "auto", F16S_GUARD),
                                               ("pretrained_ir_fp", "f16s", F16S_GUARD), ("pretrained_ir_fp", "auto", F16S_GUARD),
                                               ("pretrained_original_fp", "auto-eval", TWO_PASS_GUARD), ("pretrained_ir_fp", "auto-eval", TWO_PASS_GUARD)])
def test_released_checkpoints_load_and_match(pkg, tag, precision, tol):
    """README.md:86-95 checkpoints (as arrays): strict key match + log-probs.  "f16s" / "auto": the TRAINING
    arithmetic (eval_two_pass off: the tile-dithered single pass bench.py times); "auto-eval": what eval() runs by default (hi + lo
    split weights on every g layer).  The headline modes are held to the CONTRACT (1e-3) and, separately, to a regression guard."""
    g = gold.load(tag)
    m, meta = build_full(pkg, g, precision.split("-")[0], eval_two_pass=precision.endswith("-eval"))
    sd = {k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("sd/")}
    res = m.load_state_dict(sd, strict=False)
    assert not res.unexpected_keys and all(k.endswith("num_batches_tracked") for k in res.missing_keys), res
    m.cuda(); m.eval()
    img = torch.from_numpy(formula.hash_uniform((4, 3, 128, 128), meta["img_seed"], 0.0, 1.0)).cuda()
    qst = torch.from_numpy(formula.hash_ints((4, 20), meta["qst_seed"], 1, formula.QDICT + 1)).cuda()
    with torch.no_grad():
        lp = m(img, qst).cpu().numpy()
    e = gold.rel_err(lp, g["log_probs"])
    agree = float((lp.argmax(1) == g["log_probs"].argmax(1)).mean())
    report(tag, precision=precision, log_probs=e, argmax_agree=agree)
    if precision != "bf16" and precision != "fp32":
        assert e <= CONTRACT
    assert e <= tol, "regression guard" if precision not in ("bf16", "fp32") else "bound"
    if precision != "bf16":
        assert agree == 1.0


@pytest.mark.parametrize("precision", ["fp32", "auto", "f16s"])
def test_extraction_hooks(pkg, precision):
    """extract.py:49-74: hook on the INPUT of g_layers[2] of an ir-fp model built with extraction=True.  Under the default mode
    ("auto") and under "f16s" the hooked chain runs the parity-clean per-layer kernels: the same fp32-accurate features."""
    g = gold.load("G-extract")
    meta = g["meta"]
    hyp = dict(formula.HYP[meta["cfg"]], precision=precision)
    b, n, k, Q = meta["b"], 64, hyp["rl_in_size"] // 2, hyp["lstm_hidden"]
    rl = pkg.RelationalLayer(hyp["rl_in_size"], formula.ADICT, Q, hyp, extraction=True)
    rl.load_state_dict({k_: torch.from_numpy(v) for k_, v in formula.formula_rl_state(hyp, meta["seed"]).items()})
    rl.cuda().eval()
    got = {}

    def hook(_m, i, _o):
        feats = i[0].view(b, n * n, -1)[:, :, :-Q]
        feats = feats / feats.norm(2, 2, keepdim=True).clamp_min(1e-12)
        got["max"], got["avg"] = feats.max(1)[0].cpu().numpy(), feats.mean(1).cpu().numpy()

    rl.g_layers[meta["layer_idx"]].register_forward_hook(hook)
    x = torch.from_numpy(formula.formula_objects(b, n, k, meta["seed"] + 1)).cuda()
    assert rl(x, torch.zeros(b, Q, device="cuda")) is None
    assert gold.rel_err(got["max"], g["max"]) <= 1e-4 and gold.rel_err(got["avg"], g["avg"]) <= 1e-4


@pytest.mark.parametrize("precision,tol", [("fp32", 1e-4), ("auto", 1e-4), ("f16s", 1e-4), ("bf16", 2e-2)])
def test_extraction_native_op(pkg, precision, tol):
    """the same features from the native op (RelationalLayer.extract_features -> rn_pair_features), no hook,
    nothing materialised in fp32.  bf16 tolerance: the activations themselves carry bf16 rounding (2^-8 relative per element)."""
    g = gold.load("G-extract")
    meta = g["meta"]
    hyp = dict(formula.HYP[meta["cfg"]], precision=precision)
    b, n, k, Q = meta["b"], 64, hyp["rl_in_size"] // 2, hyp["lstm_hidden"]
    rl = pkg.RelationalLayer(hyp["rl_in_size"], formula.ADICT, Q, hyp, extraction=True)
    rl.load_state_dict({k_: torch.from_numpy(v) for k_, v in formula.formula_rl_state(hyp, meta["seed"]).items()})
    rl.cuda().eval()
    x = torch.from_numpy(formula.formula_objects(b, n, k, meta["seed"] + 1)).cuda()
    mx, av = rl.extract_features(x, torch.zeros(b, Q, device="cuda"), meta["layer_idx"])
    assert mx.shape == g["max"].shape
    assert gold.rel_err(mx.cpu().numpy(), g["max"]) <= tol and gold.rel_err(av.cpu().numpy(), g["avg"]) <= tol


@pytest.mark.parametrize("precision", ["auto", "bf16"])
@pytest.mark.parametrize("cfg", ["original-fp", "ir-fp"])
@pytest.mark.parametrize("layer_idx", [0, 1, 2, 3])
def test_extraction_fused_op_every_hook_position(pkg, cfg, layer_idx, precision):
    """SURVEY 8f row N3 as specified: RelationalLayer.extract_features -> rn_extract_features forms the input of g layer
    `layer_idx` on chip (pair rows in LDS, layers 0 .. layer_idx-1 in fp32 on the matrix pipe) -- no (B n^2, in) matrix, no stored
    activation -- for EVERY hook position of both 256-wide models, layer 0 included (F = 52: the normalised [x_j | x_i] rows), with
    a non-zero question (it enters as a per-question bias row at its injection layer).  Against the reference's own hook recipe
    (extract.py:49-74) to fp32 accuracy, whatever `precision` the module trains in (the op is fp32 by design); the hook-compatible
    path must give the same features."""
    g = gold.load("G-extract-%s-%d" % (cfg, layer_idx))
    meta = g["meta"]
    hyp = dict(formula.HYP[cfg], precision=precision)
    b, n, k, Q = meta["b"], 64, hyp["rl_in_size"] // 2, hyp["lstm_hidden"]
    rl = pkg.RelationalLayer(hyp["rl_in_size"], formula.ADICT, Q, hyp, extraction=True)
    rl.load_state_dict({k_: torch.from_numpy(v) for k_, v in formula.formula_rl_state(hyp, meta["seed"]).items()})
    rl.cuda().eval()
    x = torch.from_numpy(formula.formula_objects(b, n, k, meta["seed"] + 1)).cuda()
    q = torch.from_numpy(formula.hash_uniform((b, Q), meta["q_seed"], -1.0, 1.0)).cuda()
    calls = []
    orig = pkg.rn_hip.extract_features
    try:
        pkg.rn_hip.extract_features = lambda *a, **kw: (calls.append(1), orig(*a, **kw))[1]
        mx, av = rl.extract_features(x, q, layer_idx)
    finally:
        pkg.rn_hip.extract_features = orig
    assert calls, "the fused op was not taken"
    assert mx.shape == g["max"].shape == (b, 52 if layer_idx == 0 else 256)
    assert gold.rel_err(mx.cpu().numpy(), g["max"]) <= 2e-5 and gold.rel_err(av.cpu().numpy(), g["avg"]) <= 2e-5
    if precision == "auto":
        got = {}

        def hook(_m, i, _o):
            feats = i[0].view(b, n * n, -1)
            if layer_idx == hyp["question_injection_position"]:
                feats = feats[:, :, :-Q]
            feats = feats / feats.norm(2, 2, keepdim=True).clamp_min(1e-12)
            got["max"], got["avg"] = feats.max(1)[0].cpu().numpy(), feats.mean(1).cpu().numpy()

        hnd = rl.g_layers[layer_idx].register_forward_hook(hook)
        assert rl(x, q) is None
        hnd.remove()
        assert gold.rel_err(got["max"], g["max"]) <= 1e-4 and gold.rel_err(got["avg"], g["avg"]) <= 1e-4


def test_extraction_fused_op_ragged_object_count(pkg):
    """n = 9 objects (a 3 x 3 grid) and n = 100: tiles with fewer than 64 valid rows / two tiles per (b, i) with a ragged second
    one; against the oracle's restatement of the hook recipe."""
    from oracle import rn_oracle as O
    hyp = dict(formula.HYP["ir-fp"])
    k, Q = hyp["rl_in_size"] // 2, hyp["lstm_hidden"]
    sd = formula.formula_rl_state(hyp, 5)
    params = formula.params_from_state(sd, 4)
    rl = pkg.RelationalLayer(hyp["rl_in_size"], formula.ADICT, Q, hyp, extraction=True)
    rl.load_state_dict({k_: torch.from_numpy(v) for k_, v in sd.items()})
    rl.cuda().eval()
    for n, li in ((9, 2), (100, 1), (100, 0)):
        x = formula.formula_objects(2, n, k, 6 + n)
        q = formula.hash_uniform((2, Q), 7, -1.0, 1.0)
        mx_ref, av_ref = O.pair_features_np(x, q, params, hyp["question_injection_position"], li)
        mx, av = rl.extract_features(torch.from_numpy(x).cuda(), torch.from_numpy(q).cuda(), li)
        assert gold.rel_err(mx.cpu().numpy(), mx_ref) <= 2e-5 and gold.rel_err(av.cpu().numpy(), av_ref) <= 2e-5, (n, li)


@pytest.mark.parametrize("cfg,B,n", [("original-fp", 1, 64), ("original-fp", 3, 36), ("original-fp", 5, 49), ("original-fp", 7, 25),
                                     ("ir-fp", 2, 100), ("ir-fp", 4, 16), ("ir-fp", 3, 64), ("original-fp", 2, 9)])
@pytest.mark.parametrize("precision", ["auto", "fp32"])
def test_ragged_shapes_forward_and_backward_against_the_oracle(pkg, cfg, B, n, precision):
    """Object counts that are not the headline 64 (a 3x3 .. 10x10 grid: tiles with fewer valid rows than a wave holds, a padded j axis,
    one question per launch) and batch sizes that fill no tile: the relational layer forward + backward (nll, mean) against the
    oracle's numpy restatement of model.py:104-162 and of what autograd derives from it, on closed-form inputs and weights.
    Log-probs to the mode's bar, dx / dq / every parameter gradient in relative L2."""
    from oracle import rn_oracle as O
    hyp = dict(formula.HYP[cfg], precision=precision)
    k, Q = hyp["rl_in_size"] // 2, hyp["lstm_hidden"]
    inject = hyp["question_injection_position"]
    sd = formula.formula_rl_state(hyp, 11 + n)
    params = formula.params_from_state(sd, 4)
    rl = pkg.RelationalLayer(hyp["rl_in_size"], formula.ADICT, Q, hyp)
    rl.load_state_dict({k_: torch.from_numpy(v) for k_, v in sd.items()})
    rl.cuda().eval()
    x = formula.formula_objects(B, n, k, 3 + n)
    q = formula.hash_uniform((B, Q), 17 + B, -1.0, 1.0)
    lab = formula.hash_ints((B,), 5 + B, 0, formula.ADICT)
    lp_ref, cache = O.rl_forward_np(x, q, params, inject)
    gout = np.zeros_like(lp_ref); gout[np.arange(B), lab] = -1.0 / B
    dx_ref, dq_ref, g_ref = O.rl_backward_np(x, q, params, cache, gout)
    xt = torch.from_numpy(x).cuda().requires_grad_(True); qt = torch.from_numpy(q).cuda().requires_grad_(True)
    lp = rl(xt, qt)
    torch.nn.functional.nll_loss(lp, torch.from_numpy(lab).cuda()).backward()
    torch.cuda.synchronize()
    resolved = rl.resolved_precision(B, n, k)
    e_lp = gold.rel_err(lp.detach().cpu().numpy(), lp_ref)
    report("ragged-%s-B%d-n%d" % (cfg, B, n), precision=precision, resolved=resolved, log_probs=e_lp,
           dx_l2=l2rel(xt.grad.cpu().numpy(), dx_ref), dq_l2=l2rel(qt.grad.cpu().numpy(), dq_ref))
    exact = resolved == "fp32"
    assert e_lp <= (2e-5 if exact else 3e-4), (resolved, e_lp)
    # (fp32 path: the products are exact to fp32 accumulation order, but a pre-activation within that rounding of zero flips its
    # gate -- 3.3e-4 on dx at (B, n) = (5, 49), 5e-5 elsewhere)
    gtol = 1e-3 if exact else 4e-2
    assert l2rel(xt.grad.cpu().numpy(), dx_ref) <= gtol and l2rel(qt.grad.cpu().numpy(), dq_ref) <= gtol
    for l in range(4):
        assert l2rel(rl.g_layers[l].weight.grad.cpu().numpy(), g_ref["g_w"][l]) <= (1e-3 if exact else 4e-2), l
        assert l2rel(rl.g_layers[l].bias.grad.cpu().numpy(), g_ref["g_b"][l]) <= (1e-3 if exact else 4e-2), l
    for i, m in enumerate((rl.f_fc1, rl.f_fc2, rl.f_fc3)):
        assert l2rel(m.weight.grad.cpu().numpy(), g_ref["f_w%d" % i]) <= (1e-3 if exact else 4e-2), i
        assert l2rel(m.bias.grad.cpu().numpy(), g_ref["f_b%d" % i]) <= (1e-3 if exact else 4e-2), i


def test_changing_batch_size_and_eval_train(pkg):
    """quirk C1 fixed: a different batch size after the first forward must work."""
    hyp = dict(formula.HYP["original-fp"])

    class Args:
        qdict_size = formula.QDICT
        adict_size = formula.ADICT

    m = pkg.RN(Args, hyp)
    m.cuda()
    for b in (3, 5):
        out = m(torch.rand(b, 3, 128, 128, device="cuda"), torch.randint(1, 83, (b, 20), device="cuda"))
        assert out.shape == (b, 28) and torch.isfinite(out).all()
        assert abs(float(out.exp().sum(1).mean()) - 1.0) < 1e-4
    out.sum().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())


# ----------------------------------------------------------------------------- full BASELINE size: size-independent properties
def _full_rl(pkg, precision, seed=11):
    hyps = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "relationnetworks-clevr_amd", "config.json")))["hyperparams"]
    hyp = dict(hyps["original-fp"], precision=precision)
    torch.manual_seed(seed)
    rl = pkg.RelationalLayer(hyp["rl_in_size"], formula.ADICT, hyp["lstm_hidden"], hyp).cuda().eval()
    return rl, hyp


@pytest.mark.parametrize("precision", ["bf16", "f16s", "fp32"])
def test_full_size_properties(pkg, precision):
    """BASELINE.json configs[1] (original-fp, B=64, n=64: M = 262144 pair rows) is too large for the oracle; what the
    relation layer must satisfy at ANY size is checked there instead:
      * permutation invariance -- the answer is a sum over all object pairs (model.py:151-152): permuting a question's
        objects changes only the fp32 summation order;
      * batch independence -- question b's output and input gradients do not depend on the other questions (no
        cross-question arithmetic anywhere on the path): a 4-question slice reproduces its rows of the 64-question run;
      * the three arithmetic modes agree with each other to their documented accuracy."""
    B, n, k, Q = 64, 64, 26, 128
    rl, hyp = _full_rl(pkg, precision)
    x = torch.from_numpy(formula.hash_uniform((B, n, k), 900, -1, 1).astype(np.float32)).cuda()
    q = torch.from_numpy(formula.hash_uniform((B, Q), 901, -1, 1).astype(np.float32)).cuda()
    lab = torch.from_numpy(formula.hash_uniform((B,), 902, 0, formula.ADICT).astype(np.int64).clip(0, formula.ADICT - 1)).cuda()

    def run(xx, qq, ll):
        xx = xx.clone().requires_grad_(True); qq = qq.clone().requires_grad_(True)
        lp = rl(xx, qq)
        torch.nn.functional.nll_loss(lp, ll, reduction="sum").backward()
        return lp.detach(), xx.grad.detach(), qq.grad.detach()

    lp, dx, dq = run(x, q, lab)
    assert torch.isfinite(lp).all() and torch.isfinite(dx).all() and torch.isfinite(dq).all()
    assert torch.allclose(lp.exp().sum(1), torch.ones(B, device="cuda"), atol=1e-5)          # log-probabilities
    # permutation of the objects of every question
    perm = torch.from_numpy(np.random.RandomState(3).permutation(n)).cuda()
    lp_p, dx_p, dq_p = run(x[:, perm].contiguous(), q, lab)
    # (f16s: a pair row's position decides which of the tile-dithered weight images it multiplies -- a permutation changes more
    # than summation order, by what the mode's accuracy class allows: 2e-4)
    tol = {"fp32": 2e-6, "bf16": 2e-5}.get(precision, F16S_GUARD)
    assert gold.rel_err(lp_p.cpu().numpy(), lp.cpu().numpy()) <= tol
    assert l2rel(dx_p.cpu().numpy(), dx[:, perm].cpu().numpy()) <= (1e-5 if precision == "fp32" else 2e-2)
    assert l2rel(dq_p.cpu().numpy(), dq.cpu().numpy()) <= (1e-5 if precision == "fp32" else 2e-2)
    # a slice of the batch on its own
    sl = slice(20, 24)
    lp_s, dx_s, dq_s = run(x[sl].contiguous(), q[sl].contiguous(), lab[sl])
    assert gold.rel_err(lp_s.cpu().numpy(), lp[sl].cpu().numpy()) <= tol
    assert l2rel(dx_s.cpu().numpy(), dx[sl].cpu().numpy()) <= (1e-5 if precision == "fp32" else 2e-2)
    assert l2rel(dq_s.cpu().numpy(), dq[sl].cpu().numpy()) <= (1e-5 if precision == "fp32" else 2e-2)
    # against the exact fp32 mode of the same weights
    if precision != "fp32":
        ref, _ = _full_rl(pkg, "fp32")
        ref.load_state_dict(rl.state_dict())
        lp_r = ref(x, q).detach()
        assert gold.rel_err(lp.cpu().numpy(), lp_r.cpu().numpy()) <= (F16S_GUARD if precision == "f16s" else 1e-2)


@pytest.mark.parametrize("cfg", ["original-fp", "ir-fp"])
def test_eval_log_probs_do_not_depend_on_batch_position_or_object_order(pkg, cfg):
    """VERDICT r4 weak #3 / train.py:98-105 (the reference evaluates with whatever batch size fits): in eval() without gradients the
    chain path multiplies hi + lo split weights on every g layer (options.eval_two_pass) instead of the tile-dithered single pass --
    a question's log-probs are then the same in a batch of 64, alone in a slice of 4 at another batch position, and for any order
    of its objects (<= 5e-5; measured ~1e-6: fp32 summation order), and within TWO_PASS_GUARD of the exact fp32 mode.  With the
    option off eval() runs the training arithmetic, whose weight image depends on the pair row's tile (<= F16S_GUARD)."""
    B, n, k, Q = 64, 64, 26, 128
    hyp = dict(formula.HYP[cfg], precision="auto")
    torch.manual_seed(11)
    rl = pkg.RelationalLayer(hyp["rl_in_size"], formula.ADICT, hyp["lstm_hidden"], hyp).cuda().eval()
    assert rl.eval_two_pass is True
    x = torch.from_numpy(formula.hash_uniform((B, n, k), 900, -1, 1).astype(np.float32)).cuda()
    q = torch.from_numpy(formula.hash_uniform((B, Q), 901, -1, 1).astype(np.float32)).cuda()
    perm = torch.from_numpy(np.random.RandomState(3).permutation(n)).cuda()
    errs = {}
    for two in (True, False):
        rl.eval_two_pass = two
        with torch.no_grad():
            lp = rl(x, q)
            lp_s = rl(x[20:24].contiguous(), q[20:24].contiguous())
            lp_o = rl(x[1:5].contiguous(), q[1:5].contiguous())              # (an odd tile offset: question 1 starts at tile 16)
            lp_p = rl(x[:, perm].contiguous(), q)
        errs[two] = (gold.rel_err(lp_s.cpu().numpy(), lp[20:24].cpu().numpy()), gold.rel_err(lp_o.cpu().numpy(), lp[1:5].cpu().numpy()),
                     gold.rel_err(lp_p.cpu().numpy(), lp.cpu().numpy()))
        if two:
            lp2 = lp
    ref = pkg.RelationalLayer(hyp["rl_in_size"], formula.ADICT, hyp["lstm_hidden"], dict(hyp, precision="fp32")).cuda().eval()
    ref.load_state_dict(rl.state_dict())
    with torch.no_grad():
        e_ref = gold.rel_err(lp2.cpu().numpy(), ref(x, q).cpu().numpy())
    report("eval-invariance-" + cfg, precision="auto", two_pass=errs[True], training_arithmetic=errs[False], two_pass_vs_fp32=e_ref)
    assert max(errs[True]) <= 5e-5, errs
    assert max(errs[False]) <= F16S_GUARD, errs
    assert e_ref <= TWO_PASS_GUARD
    # a forward pass that needs a gradient runs the training arithmetic whatever the mode (the backward needs its masks / copies)
    rl.eval_two_pass = True
    xg = x[:4].clone().requires_grad_(True)
    lp_g = rl(xg, q[:4].contiguous())
    rl.train(); rl.forced_dropout_mask = torch.ones(4, rl.f_fc2.out_features, device="cuda")
    lp_t = rl(xg, q[:4].contiguous())
    assert torch.equal(lp_g, lp_t)


# ----------------------------------------------------------------------------- BASELINE configs[4] at its real size
@pytest.mark.parametrize("precision", ["auto", "bf16", "f16s", "fp32"])
def test_stress_config_real_dispatch(pkg, precision):
    """original-fp stress: 14x14 grid (n = 196), B = 32 -- M = 1,229,312 pair rows, n % 32 != 0: "auto" / "f16s" run the factored
    register-resident chain on the PADDED j axis (196 -> 224 pair rows per (question, i) group, 256-row tiles that may straddle
    two questions, two partial pair-sum rows per tile; no pair matrix, no stored H_3 / dZ_3); "bf16" / "fp32" the per-layer
    kernels on the materialised pair matrix.  Checked against G-fp196-b32, recorded
    from the reference at this very size (make_golden.py stress): log-probs, loss, dq, dx (norm + 4096 sampled entries),
    bias gradients, weight-gradient norms + samples; plus the size-independent properties (object permutation, batch slice)."""
    g = gold.load("G-fp196-b32")
    meta = g["meta"]
    assert meta["b"] == 32 and meta["n"] == 196
    hyp, sd, x, q, lab = gold.rl_case(meta)
    rl = pkg.RelationalLayer(hyp["rl_in_size"], formula.ADICT, hyp["lstm_hidden"], dict(hyp, precision=precision))
    rl.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    rl = rl.cuda().eval()
    labt = torch.from_numpy(lab).cuda()

    def run(xn, qn, ll):
        xt = torch.from_numpy(xn).cuda().requires_grad_(True)
        qt = torch.from_numpy(qn).cuda().requires_grad_(True)
        lp = rl(xt, qt)
        loss = torch.nn.functional.nll_loss(lp, ll)
        loss.backward()
        torch.cuda.synchronize()
        return lp.detach().cpu().numpy(), float(loss.detach()), xt.grad.cpu().numpy(), qt.grad.cpu().numpy()

    for p_ in rl.parameters():
        p_.grad = None
    lp, loss, dx, dq = run(x, q, labt)
    grads = {n_: p_.grad.detach().cpu().numpy() for n_, p_ in rl.named_parameters()}
    e_lp = gold.rel_err(lp, g["log_probs"])
    e_dq = l2rel(dq, g["dq"])
    e_dxs = l2rel(dx.reshape(-1)[g["dx_sample_idx"]], g["dx_sample"])
    e_dxn = abs(float(np.linalg.norm(dx.astype(np.float64))) - float(g["dx_norm"])) / float(g["dx_norm"])
    e_b = max(l2rel(grads[k_[5:]], g[k_]) for k_ in g if k_.startswith("grad/"))
    e_wn = max(abs(float(np.linalg.norm(grads[k_[9:]].astype(np.float64))) - float(g[k_])) / max(float(g[k_]), 1e-30)
               for k_ in g if k_.startswith("gradnorm/"))
    agree = float((lp.argmax(1) == g["log_probs"].argmax(1)).mean())
    report("G-fp196-b32", precision=precision, resolved=rl.resolved_precision(32, 196, 26), log_probs=e_lp, loss=abs(loss - float(g["loss"])) / float(g["loss"]),
           dq_l2=e_dq, dx_sample_l2=e_dxs, dx_norm=e_dxn, bias_l2=e_b, wnorm=e_wn, argmax_agree=agree)
    lp_tol, g_tol = {"fp32": (1e-5, 1e-3), "f16s": (F16S_GUARD, 1.2e-2), "auto": (F16S_GUARD, 1.2e-2), "bf16": (3e-3, 6e-2)}[precision]
    assert np.isfinite(lp).all() and e_lp <= lp_tol
    assert abs(loss - float(g["loss"])) <= max(lp_tol, 1e-6) * 10 * abs(float(g["loss"]))
    assert e_dq <= g_tol and e_dxs <= g_tol and e_dxn <= g_tol and e_b <= g_tol and e_wn <= g_tol
    if precision != "bf16":
        assert agree == 1.0
    # size-independent properties at the full size: a slice of the batch on its own reproduces its rows (8 questions:
    # M = 8 * 196^2 stays a multiple of 128, so that "auto" / "f16s" run the same kernels on the slice)
    sl = slice(8, 16)
    lp_s, _, dx_s, dq_s = run(np.ascontiguousarray(x[sl]), np.ascontiguousarray(q[sl]), labt[sl])
    ptol = {"fp32": 2e-6, "bf16": 5e-5}.get(precision, F16S_GUARD)     # (f16s / auto: the tile-dithered weight image depends on the row's position)
    assert gold.rel_err(lp_s, lp[sl]) <= ptol
    gt = 1e-5 if precision == "fp32" else 3e-2
    assert l2rel(dx_s * (8 / 32), dx[sl]) <= gt and l2rel(dq_s * (8 / 32), dq[sl]) <= gt       # (mean-loss scaling)
    # ... and permuting every question's objects changes only summation order
    perm = np.random.RandomState(4).permutation(196)
    lp_p, _, dx_p, dq_p = run(np.ascontiguousarray(x[:, perm]), q, labt)
    assert gold.rel_err(lp_p, lp) <= ptol
    assert l2rel(dx_p, dx[:, perm]) <= gt and l2rel(dq_p, dq) <= gt


@pytest.mark.parametrize("cfg", ["original-fp", "ir-fp"])
@pytest.mark.parametrize("precision", ["auto"])
def test_fused_coordinate_tagging_equals_the_concatenated_path(pkg, cfg, precision):
    """RN.forward hands the conv grid and the (2, n) coordinate table to the kernels separately (rn_pair_tables /
    rn_wgrad0_from_reductions tag the coordinates themselves, rn_pair_dx_dq writes the gradient in the grid's layout); with
    that path disabled it concatenates like the reference (model.py:195-201).  Same arithmetic on the same values: log-probs,
    loss and EVERY gradient must be bitwise equal."""
    class Args:
        qdict_size = formula.QDICT
        adict_size = formula.ADICT

    img = torch.from_numpy(formula.hash_uniform((8, 3, 128, 128), 321, 0.0, 1.0)).cuda()
    qst = torch.from_numpy(formula.hash_ints((8, 20), 322, 1, formula.QDICT + 1)).cuda()
    lab = torch.from_numpy(formula.hash_ints((8,), 323, 0, formula.ADICT)).cuda()

    def run(fast):
        torch.manual_seed(9)
        m = pkg.RN(Args, dict(formula.HYP[cfg], precision=precision, dropout=0.0)).cuda()
        m.train()
        assert m.rl.grid_fast_path(8, 64, 26)
        if not fast:                                          # the reference's own sequence: concatenate, then the layer (model.py:195-204)
            m.rl.grid_fast_path = lambda *a_, **k_: False
        lp, loss = m.forward_loss(img, qst, lab)
        loss.backward()
        torch.cuda.synchronize()
        return lp.detach().clone(), float(loss.detach()), {n_: p_.grad.clone() for n_, p_ in m.named_parameters()}

    lp_a, loss_a, g_a = run(True)
    lp_b, loss_b, g_b = run(False)
    assert torch.equal(lp_a, lp_b) and loss_a == loss_b
    for n_ in g_b:
        assert torch.equal(g_a[n_], g_b[n_]), n_


@pytest.mark.gpu
@pytest.mark.parametrize("B", [8, 64])
def test_question_gradient_hand_off_by_event_matches_the_synchronous_path(pkg, B):
    """ir-fp (question injected at layer 2, model.py:131-142): the question gradient is formed on the weight-gradient stream from the
    per-question sums of the stored dZ_2 and handed to the question encoder's backward BY EVENT (functional._GRAD_EVENTS) -- the main
    stream goes on with dx.  Against the synchronous hand-off (SCHED dq_async = 0: dq from the weight-gradient launch's per-question
    partials where its splits are question-aligned, B = 64; from the same sums on the main stream otherwise, B = 8): same log-probs
    and loss bitwise, every gradient to fp32 summation order -- the encoder's own gradients included, which is what would be garbage
    if its backward ran ahead of the event.  Three passes each: the hand-off registry must not leak between passes."""
    class Args:
        qdict_size = formula.QDICT
        adict_size = formula.ADICT

    RF = pkg.functional
    img = torch.from_numpy(formula.hash_uniform((B, 3, 128, 128), 331, 0.0, 1.0)).cuda()
    qst = torch.from_numpy(formula.hash_ints((B, 20), 332, 1, formula.QDICT + 1)).cuda()
    lab = torch.from_numpy(formula.hash_ints((B,), 333, 0, formula.ADICT)).cuda()

    def run(async_):
        old = RF.SCHED["dq_async"]
        RF.SCHED["dq_async"] = async_
        try:
            torch.manual_seed(11)
            m = pkg.RN(Args, dict(formula.HYP["ir-fp"], precision="auto", dropout=0.0)).cuda()
            m.train()
            outs = []
            for _ in range(3):
                m.zero_grad(set_to_none=True)
                lp, loss = m.forward_loss(img, qst, lab)
                assert not m.rl._packed.q_grad_async                  # (the permission is one call's: consumed by the layer's forward)
                n0 = RF.HANDED_BY_EVENT[0]
                loss.backward()
                torch.cuda.synchronize()
                assert RF.HANDED_BY_EVENT[0] - n0 == (1 if async_ else 0)   # the event path ran / did not run
                assert not RF._GRAD_EVENTS                            # ... and the encoder's backward consumed the entry
                outs.append((lp.detach().clone(), float(loss.detach()), {n_: p_.grad.clone() for n_, p_ in m.named_parameters()}))
            return outs
        finally:
            RF.SCHED["dq_async"] = old

    a, s = run(1), run(0)
    for (lp_a, loss_a, g_a), (lp_s, loss_s, g_s) in zip(a, s):
        assert torch.equal(lp_a, lp_s) and loss_a == loss_s
        for n_ in g_s:
            assert l2rel(g_a[n_].cpu().numpy(), g_s[n_].cpu().numpy()) <= 2e-5, n_
    # a relational layer called with a question of the caller's own gets the synchronous hand-off (no permission was given)
    m = pkg.RN(Args, dict(formula.HYP["ir-fp"], precision="auto", dropout=0.0)).cuda().train()
    x = torch.from_numpy(formula.hash_uniform((B, 64, 26), 334, -1.0, 1.0)).cuda().requires_grad_()
    q = torch.from_numpy(formula.hash_uniform((B, 128), 335, -1.0, 1.0)).cuda().requires_grad_()
    n0 = RF.HANDED_BY_EVENT[0]
    m.rl(x, q).sum().backward()
    torch.cuda.synchronize()
    assert RF.HANDED_BY_EVENT[0] == n0 and q.grad is not None and torch.isfinite(q.grad).all()
    for n_ in a[0][2]:                                                # ... and a pass repeats itself bitwise
        assert torch.equal(a[0][2][n_], a[2][2][n_]), n_
    assert float(a[0][2]["text.wembedding.weight"].abs().sum()) > 0
