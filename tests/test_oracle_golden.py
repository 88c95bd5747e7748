"""CPU: pin oracle/ against the golden vectors recorded from the reference."""
import numpy as np
import pytest
import torch

import gold
from oracle import formula, rn_oracle as O

RL_TAGS = ["G-sd4", "G-irsd4", "G-fp-small", "G-ir-small", "G-fp64", "G-ir64", "G-fp196", "G-drop"]
TOL = 2e-5   # fp32 CPU restatement vs fp32 reference: only summation-order noise


@pytest.mark.parametrize("tag", RL_TAGS)
def test_numpy_restatement_matches_reference(tag):
    g = gold.load(tag)
    hyp, sd, x, q, lab = gold.rl_case(g["meta"])
    params = formula.params_from_state(sd, len(hyp["g_layers"]))
    mask = g.get("dropout_mask")
    lp, cache = O.rl_forward_np(x, q, params, hyp["question_injection_position"], dropout_mask=mask)
    assert gold.rel_err(lp, g["log_probs"]) <= TOL
    assert gold.rel_err(cache["x_g"], g["x_g"]) <= TOL
    loss = -lp[np.arange(len(lab)), lab].mean()
    assert abs(loss - float(g["loss"])) <= TOL * max(1.0, abs(float(g["loss"])))
    if "pair_rows_b0" in g:
        P = O.pair_matrix(x, None)
        rows = g["pair_rows_b0"]
        assert np.array_equal(P[: rows.shape[0]], rows)          # exact: pure data movement
    if "inj_in_rows" in g:
        inj = hyp["question_injection_position"]
        assert gold.rel_err(cache["acts"][inj][g["inj_rows"]], g["inj_in_rows"]) <= TOL
    dx, dq, gr = O.rl_backward_np(x, q, params, cache, gold.nll_grad(lp, lab))
    assert gold.rel_err(dx, g["dx"]) <= 5 * TOL
    assert gold.rel_err(dq, g["dq"]) <= 5 * TOL
    named = {}
    for i in range(len(hyp["g_layers"])):
        named["g_layers.%d.weight" % i] = gr["g_w"][i]; named["g_layers.%d.bias" % i] = gr["g_b"][i]
    for i in range(3):
        named["f_fc%d.weight" % (i + 1)] = gr["f_w%d" % i]; named["f_fc%d.bias" % (i + 1)] = gr["f_b%d" % i]
    gold.check_grads(g, named, 10 * TOL)


@pytest.mark.parametrize("tag", ["G-sd4", "G-fp-small", "G-ir-small", "G-drop"])
def test_torch_restatement_matches_reference(tag):
    g = gold.load(tag)
    hyp, sd, x, q, lab = gold.rl_case(g["meta"])
    rl = O.RelationalLayerOracle(hyp["rl_in_size"], formula.ADICT, hyp["lstm_hidden"], hyp)
    rl.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    rl.eval()
    if "dropout_mask" in g:
        rl.forced_dropout_mask = torch.from_numpy(g["dropout_mask"])
    xt = torch.from_numpy(x).requires_grad_(True); qt = torch.from_numpy(q).requires_grad_(True)
    lp = rl(xt, qt)
    torch.nn.functional.nll_loss(lp, torch.from_numpy(lab)).backward()
    assert gold.rel_err(lp.detach().numpy(), g["log_probs"]) <= TOL
    assert gold.rel_err(xt.grad.numpy(), g["dx"]) <= 5 * TOL
    assert gold.rel_err(qt.grad.numpy(), g["dq"]) <= 5 * TOL
    gold.check_grads(g, {n: p.grad.numpy() for n, p in rl.named_parameters()}, 10 * TOL)


def test_coord_table_bit_exact_vs_torch():
    for d in (1, 2, 7, 8, 14, 16):
        assert np.array_equal(O.coord_table(d), torch.linspace(-d / 2.0, d / 2.0, d).numpy()), d
    obj = O.grid_to_objects(np.zeros((1, 24, 8, 8), dtype=np.float32))
    assert obj.shape == (1, 64, 26)
    assert abs(obj[0, 3, 24] - (-0.5714)) < 1e-3 and obj[0, 3, 25] == -4.0     # SURVEY 8a row a3 probe


@pytest.mark.parametrize("tag", ["G-e2e", "G-e2e-ir"])
def test_full_model_restatement(tag):
    g = gold.load(tag)
    meta = g["meta"]
    hyp = formula.HYP[meta["cfg"]]
    m = O.RNOracle(formula.QDICT, formula.ADICT, hyp)
    shapes = {k: tuple(v) for k, v in __import__("json").loads(str(g["state_names"])).items()}
    sd = formula.formula_fill_state(shapes, meta["seed"])
    missing = m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    m.eval()
    img = torch.from_numpy(formula.hash_uniform((meta["b"], 3, meta["img_hw"], meta["img_hw"]), meta["seed"] + 1, 0.0, 1.0))
    qst = torch.from_numpy(formula.hash_ints((meta["b"], meta["T"]), meta["seed"] + 2, 1, formula.QDICT + 1))
    with torch.no_grad():
        lp = m(img, qst)
    assert gold.rel_err(lp.numpy(), g["log_probs"]) <= TOL


@pytest.mark.parametrize("cfg", ["original-fp", "ir-fp"])
@pytest.mark.parametrize("layer_idx", [0, 1, 2, 3])
def test_extraction_restatement(cfg, layer_idx):
    """oracle.pair_features_np against the features the REFERENCE's hook recipe (extract.py:49-74) produced for every hook
    position of both 256-wide models, with a non-zero question (tests/golden/make_golden.py: record_extract)."""
    g = gold.load("G-extract-%s-%d" % (cfg, layer_idx))
    meta = g["meta"]
    hyp = formula.HYP[cfg]
    b, n, k, Q = meta["b"], 64, hyp["rl_in_size"] // 2, hyp["lstm_hidden"]
    params = formula.params_from_state(formula.formula_rl_state(hyp, meta["seed"]), 4)
    x = formula.formula_objects(b, n, k, meta["seed"] + 1)
    q = formula.hash_uniform((b, Q), meta["q_seed"], -1.0, 1.0)
    mx, av = O.pair_features_np(x, q, params, hyp["question_injection_position"], layer_idx)
    assert mx.shape == g["max"].shape
    assert gold.rel_err(mx, g["max"]) <= TOL and gold.rel_err(av, g["avg"]) <= TOL
