"""CPU (no GPU): the C-ABI library loads and exports every symbol include/rn_hip.h declares, the
host-side mirror of the reference interface (class names, state_dict keys, shape planning) is
right, and the product path refuses to run without the GPU (no silent CPU fallback)."""
import ctypes
import json
import os
import re

import numpy as np
import pytest
import torch

from oracle import formula

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


@pytest.fixture(scope="module")
def pkg():
    import __graft_entry__ as ge
    import relationnetworks_clevr_amd as p
    if not os.path.exists(p.rn_hip.LIB_PATH):
        ge.build()
    return p


def test_library_exports_every_declared_symbol(pkg):
    hdr = open(os.path.join(ROOT, "include", "rn_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(rn_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 18
    lib = ctypes.CDLL(pkg.rn_hip.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), "librn_hip.so does not export %s" % name
    assert declared == set(pkg.rn_hip.SIGNATURES), declared ^ set(pkg.rn_hip.SIGNATURES)
    # diagnostics live in their own header, outside the product ABI
    dbg = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "rn_hip_debug.h")).read(), flags=re.S)
    dbg_declared = set(re.findall(r"\b(rn_[a-z0-9_]+)\s*\(", dbg))
    assert dbg_declared == set(pkg.rn_hip.DEBUG_SIGNATURES) and not (dbg_declared & declared)
    for name in sorted(dbg_declared):
        assert hasattr(lib, name), "librn_hip.so does not export %s" % name
    loaded = pkg.rn_hip.load()
    # one version number in three places: the header, the library, the binding (bumped whenever a signature or a layout changes)
    hv = int(re.search(r"#define\s+RN_ABI_VERSION\s+(\d+)", open(os.path.join(ROOT, "include", "rn_hip.h")).read()).group(1))
    assert loaded.rn_abi_version() == hv == pkg.rn_hip.ABI_VERSION
    # pure host entry points (no device work) are callable without a GPU
    Hm = pkg.rn_hip
    ws = Hm.workspace_bytes
    assert ws(Hm.WS_WGRAD, 262144, 256, 256) == (256 * 256 * 256 + 256 * 256) * 4
    assert ws(Hm.WS_WGRAD, 100, 100, 256) == 0
    assert ws(Hm.WS_PAIR_SUM, 64, 4096, 256) == 64 * 16 * 256 * 4
    assert ws(Hm.WS_RR_MASK, 262144) == 32 * 262144 and ws(Hm.WS_EXTRACT, 64, 64, 256) == 2 * 64 * 64 * 256 * 4
    assert ws(99) == 0 and b"unknown op" in loaded.rn_last_error()
    # the Python constants are the header's enum
    hdr = open(os.path.join(ROOT, "include", "rn_hip.h")).read()
    for name, val in re.findall(r"(RN_WS_[A-Z0-9_]+) = (\d+)", hdr):
        assert getattr(Hm, name[3:]) == int(val), name
    assert len(declared) <= 61
    # row splits of the blocked weight gradient: a budget of 4 x 40 workgroups (round 6: 160 of the 256 CUs; rounds 4-5: 192); an
    # `aligned` launch: njobs x Z x 4 workgroups, Z question-aligned
    sp = loaded.rn_wgrad_blocked_splits
    assert sp(64 * 4096, 4096, 1, 0) == 40 and sp(64 * 4096, 4096, 3, 0) == 13 and sp(2 * 1024, 1024, 1, 0) == 32 and sp(100, 0, 1, 0) == 0
    assert sp(64 * 4096, 4096, 1, 1) == 64 and sp(32 * 4096, 4096, 1, 1) == 32 and sp(3 * 4096, 4096, 1, 1) == 24 and sp(128 * 4096, 4096, 1, 1) == 128
    assert sp(17 * 4096, 4096, 1, 1) == 34 and sp(32 * 38416, 38416, 1, 1) == 40 and sp(64 * 4096, 4096, 3, 1) == 64 and sp(4 * 4096, 4096, 3, 1) == 8
    assert sp(64 * 4096, 4096, 5, 0) == 0
    # the product launch (not aligned, e4m3 images): wide units for the stored jobs, quad units for the gate job, 160 workgroups in all
    def mix(M, nw, nq):
        zw, zq = ctypes.c_int(-1), ctypes.c_int(-1)
        assert loaded.rn_debug_wgrad_blocked_mix(M, nw, nq, ctypes.byref(zw), ctypes.byref(zq)) == 0
        return zw.value, zq.value
    assert mix(64 * 4096, 2, 1) == (48, 16) and 2 * 48 + 4 * 16 == 160        # the step's launch: two stored gradients + the gate job
    assert mix(64 * 4096, 1, 0) == (160, 0) and mix(64 * 4096, 0, 1) == (0, 40) and mix(64 * 4096, 3, 0) == (53, 0)
    assert mix(1024, 2, 1) == (16, 16)                                         # (never more splits than 64-row steps)
    for M_, nw_, nq_ in ((640 * 4096, 2, 1), (32 * 196 * 224, 2, 1), (17 * 4096, 1, 1)):
        zw, zq = mix(M_, nw_, nq_)
        assert nw_ * zw + 4 * nq_ * zq <= 160 and zw >= 2 * zq > 0, (M_, zw, zq)
    # workspace: one fp32 256 x 256 partial tile + 4 db rows per row split; not aligned: the most splits any mix of wide (one
    # workgroup per split) and quad (four) jobs gets out of the 160 workgroups -- three wide jobs: 3 x 53; aligned: uniform splits
    assert ws(Hm.WS_WGRAD_BLOCKED, 64 * 4096, 4096, 3, 0) == 159 * (65536 + 4 * 256) * 4
    assert ws(Hm.WS_WGRAD_BLOCKED, 64 * 4096, 4096, 3, 1) == 3 * 64 * (65536 + 4 * 256) * 4
    assert ws(Hm.WS_WGRAD_BLOCKED, 2 * 1024, 1024, 1, 0) == 32 * (65536 + 4 * 256) * 4        # (never more splits than 64-row steps)

def test_library_and_hot_path_never_read_the_environment(pkg):
    """dispatch must not depend on the process environment at call time.  The product librn_hip.so does not even
    import getenv (kernel-variant knobs and timing ablations exist in RN_DIAG builds only, behind rn_diag_env); the Python side
    reads the RN_* variables ONCE, at import, into options.OPT."""
    import subprocess
    und = subprocess.run(["nm", "-D", "--undefined-only", pkg.rn_hip.LIB_PATH], capture_output=True, text=True).stdout
    assert "getenv" not in und
    csrc = os.path.join(ROOT, "relationnetworks-clevr_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        src = open(os.path.join(csrc, f)).read()
        if f == "rn_common.h":
            assert src.count("getenv(") == 1 and "#ifdef RN_DIAG" in src            # the one call, inside the diagnostics helper
        else:
            assert "getenv(" not in src, f
    pk = os.path.join(ROOT, "relationnetworks-clevr_amd")
    for f in sorted(os.listdir(pk)):
        if f.endswith(".py") and f not in ("options.py", "_build.py", "train.py"):   # (train.py: torchrun's RANK / WORLD_SIZE in main())
            assert "environ" not in open(os.path.join(pk, f)).read(), f
    O = pkg.options
    assert O.OPT.h8 is True and O.OPT.precision == "auto" and O.OPT.non_default() == {}
    o2 = O.Options({"RN_H8": "0", "RN_NO_CHAIN_REDUCE": "1", "RN_PRECISION": "fp32", "RN_NO_FUSED_ADAM": "0", "RN_OVERLAP_STREAMS": "0"})
    assert (o2.h8, o2.chain_reduce, o2.precision, o2.fused_adam, o2.overlap_streams) == (False, False, "fp32", True, False)
    assert len(O._SPEC) <= 15
    with O.override(h8=False):
        assert O.OPT.h8 is False
    assert O.OPT.h8 is True
    with pytest.raises(AttributeError):
        with O.override(no_such_option=1):
            pass


def test_argument_validation_without_gpu(pkg):
    lib = pkg.rn_hip.load()
    rc = lib.rn_pair_build_fwd(None, 0, 0, 0, None, 0, None, 0, 1, 1, 1, 0, 64, None)
    assert rc < 0 and b"bad pointer" in lib.rn_last_error()
    rc = lib.rn_g_linear_fwd(1 << 20, 256, 1 << 20, 256, 1 << 20, 1 << 20, 256, 0, 128, 100, 256, None)
    assert rc < 0 and b"multiple of 64" in lib.rn_last_error()


def test_missing_library_is_loud(pkg, tmp_path):
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        pkg.rn_hip.load(str(tmp_path / "nope.so"))


@pytest.mark.parametrize("cfg", ["original-fp", "original-sd", "ir-fp", "ir-sd"])
def test_module_surface_matches_reference_contract(pkg, cfg):
    """names, shapes and creation order of the parameters; attributes callers touch."""
    hyps = json.load(open(os.path.join(ROOT, "relationnetworks-clevr_amd", "config.json")))["hyperparams"]
    assert hyps == formula.HYP
    hyp = hyps[cfg]

    class Args:
        qdict_size, adict_size = formula.QDICT, formula.ADICT

    m = pkg.RN(Args, hyp)
    keys = list(m.state_dict().keys())
    expect = []
    for i in range(1, 5):
        expect += ["conv.conv%d.weight" % i, "conv.conv%d.bias" % i]
        expect += ["conv.batchNorm%d.%s" % (i, s) for s in ("weight", "bias", "running_mean", "running_var", "num_batches_tracked")]
    expect += ["text.wembedding.weight"] + ["text.lstm.%s_l0" % s for s in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
    expect += ["rl.f_fc%d.%s" % (i, s) for i in (1, 2, 3) for s in ("weight", "bias")]
    expect += ["rl.g_layers.%d.%s" % (i, s) for i in range(4) for s in ("weight", "bias")]
    assert keys == expect
    for name, (o, i) in formula.rl_layer_shapes(hyp):
        assert tuple(m.state_dict()["rl." + name + ".weight"].shape) == (o, i)
    for attr in ("conv", "text", "rl", "coord_tensor", "on_gpu", "state_desc"):
        assert hasattr(m, attr)
    assert isinstance(m.rl.g_layers, torch.nn.ModuleList) and isinstance(m.rl.dropout, torch.nn.Dropout)
    assert m.rl.dropout.p == hyp["dropout"] and m.rl.quest_inject_position == hyp["question_injection_position"]
    assert m.text.wembedding.num_embeddings == formula.QDICT + 1
    assert all(p.dtype == torch.float32 for p in m.parameters())
    n_par = sum(p.numel() for p in m.parameters())
    assert n_par == (484580 if cfg.endswith("fp") else 2059492)


def test_released_checkpoints_load_strictly(pkg):
    for tag, cfg in (("pretrained_original_fp", "original-fp"), ("pretrained_ir_fp", "ir-fp")):
        z = np.load(os.path.join(ROOT, "tests", "golden", tag + ".npz"))

        class Args:
            qdict_size, adict_size = formula.QDICT, formula.ADICT

        m = pkg.RN(Args, formula.HYP[cfg])
        sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd/")}
        assert len(sd) == 43
        res = m.load_state_dict(sd, strict=False)
        assert not res.unexpected_keys and all(k.endswith("num_batches_tracked") for k in res.missing_keys)
        # DataParallel-style 'module.' prefix (train.py:271-274) round trip
        pref = {"module." + k: v for k, v in m.state_dict().items()}
        m.load_state_dict({k[len("module."):]: v for k, v in pref.items()}, strict=True)


def test_no_cpu_fallback(pkg):
    hyp = formula.HYP["original-sd"]
    rl = pkg.RelationalLayer(hyp["rl_in_size"], 28, hyp["lstm_hidden"], hyp)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        rl(torch.zeros(2, 12, 7), torch.zeros(2, 256))


def test_layer_plan_shapes(pkg):
    RF = pkg.functional
    p = RF.LayerPlan(26, 128, [256] * 4, 0)
    assert p.ktrue == [180, 256, 256, 256] and p.kpad == [192, 256, 256, 256]
    p = RF.LayerPlan(26, 128, [256] * 4, 2)
    assert p.ktrue == [52, 256, 384, 256] and p.kpad == [64, 256, 384, 256]
    p = RF.LayerPlan(7, 256, [512] * 4, 0)
    assert p.ktrue == [270, 512, 512, 512] and p.kpad == [320, 512, 512, 512]
    with pytest.raises(RuntimeError, match="multiples of 256"):
        RF.LayerPlan(7, 256, [100, 100], 0)


def test_product_never_imports_the_oracle():
    """③: only tests/, smoke() and bench.py's cpu_baseline may touch oracle/."""
    pdir = os.path.join(ROOT, "relationnetworks-clevr_amd")
    for dp, _dn, fn in os.walk(pdir):
        for f in fn:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dp, f)).read()
                assert "oracle" not in src.replace("the oracle", ""), f
                # the reference may be CITED in docstrings / comments, never touched by code
                code = re.sub(r'"""[\s\S]*?"""', "", src)
                code = "\n".join(l for l in code.splitlines() if not l.strip().startswith(("#", "//", "*", "/*")))
                assert "/root/reference" not in code, f


def test_flat_import_like_the_reference_has_no_relative_imports_behind_it():
    """train.py run like the reference's (the package directory itself on sys.path: `import model`, `import dp`, train.py:25) must
    not meet a relative import later, inside a function -- ADVICE r3: DataParallelTrainer.step() did `from . import rn_hip`, which
    raises 'attempted relative import with no known parent package' on the first captured step.  (a) no function body of dp.py /
    train.py holds a relative import; (b) the flat imports work in a fresh interpreter and share ONE binding module with
    `functional` (dp reaches the library through RF.H)."""
    import ast
    import subprocess
    import sys
    pkgdir = os.path.join(ROOT, "relationnetworks-clevr_amd")
    for name in ("dp.py", "train.py"):
        tree = ast.parse(open(os.path.join(pkgdir, name)).read())
        for fn in ast.walk(tree):
            if isinstance(fn, (ast.FunctionDef, ast.AsyncFunctionDef)):
                for node in ast.walk(fn):
                    assert not (isinstance(node, ast.ImportFrom) and node.level > 0), "%s: relative import inside %s()" % (name, fn.name)
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r); import dp, train; "
            "assert dp.RF.H is sys.modules['relationnetworks_clevr_amd.rn_hip'] or dp.RF.H.__name__.endswith('rn_hip'); print('flat ok')"
            % (ROOT, pkgdir))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=pkgdir)
    assert r.returncode == 0 and "flat ok" in r.stdout, r.stderr[-2000:]


@pytest.mark.parametrize("B,n,tpu,whole", [(32, 196, 5, "lib"), (32, 196, 5, 0), (32, 196, 5, "all"), (32, 196, 1, "lib"), (64, 64, 4, "lib"), (17, 64, 1, 100),
                                          (3, 64, 8, 5), (3, 64, 4, 0), (2, 96, 3, 7), (4, 40, 5, 3), (2, 196, 25, 9), (40, 64, 2, "lib")])
def test_reducing_chain_schedule_covers_every_tile_and_record_once(pkg, B, n, tpu, whole):
    """Host logic of the reducing backward chain's BALANCED schedule (rn_common.h rn_red_item / rn_red_walk_pos, walked through
    rn_probe_red_schedule -- no GPU): every tile of the launch is run by exactly one work item; a whole unit is tiles_per_unit
    consecutive tiles of ONE (question, j block) and leaves the unit's own record; the tail's single tiles leave one record each,
    all distinct and inside the buffer; and the records rn_pair_reduce_parts adds up for a unit -- its own + the extra ones at the
    unit's walk position (question fastest) -- are exactly those written for that unit's tiles.  (32, 196, 5): BASELINE.json
    configs[4] on a 256-CU chip: 1024 whole units + 480 single tiles, three tail units per question."""
    lib = pkg.rn_hip.load()
    njp = (n + 31) // 32 * 32
    M = B * n * njp
    tpbj, jgs = (n + 7) // 8, njp // 32
    assert tpbj % tpu == 0 and lib.rn_g_chain_bwd_rr_red_tpu(M, n, njp) > 0
    nu = tpbj // tpu
    nunits = B * jgs * nu
    ntiles = nunits * tpu
    lib_whole = lib.rn_g_chain_bwd_rr_red_whole(M, n, njp, tpu)
    assert 0 <= lib_whole <= nunits and (tpu == 1 and lib_whole == nunits or tpu > 1 and lib_whole % 256 == 0 and nunits - lib_whole < 256)
    whole = {"lib": lib_whole, "all": nunits}.get(whole, whole)
    nitems = whole + (nunits - whole) * tpu
    out = (ctypes.c_int * (3 * nitems))()
    assert lib.rn_probe_red_schedule(M, n, njp, tpu, whole, out, nitems) == nitems
    assert lib.rn_probe_red_schedule(M, n, njp, tpu, whole, out, nitems - 1) < 0          # (too small a buffer is refused)
    assert lib.rn_probe_red_schedule(M, n, njp, tpu, nunits + 1, out, nitems) < 0
    items = np.frombuffer(out, dtype=np.int32).reshape(nitems, 3)
    records = nunits + (nunits - whole) * (tpu - 1)
    assert records == pkg.rn_hip.g_chain_bwd_rr_red_records(M, n, njp, tpu, whole)
    seen_tiles = np.zeros(ntiles, dtype=np.int32)
    written = {}                                            # record -> unit whose tiles it sums
    for i, (tile0, tcount, rec) in enumerate(items):
        assert tcount == (tpu if i < whole else 1)
        seen_tiles[tile0:tile0 + tcount] += 1
        unit = tile0 // tpu
        assert (tile0 + tcount - 1) // tpu == unit          # an item never straddles two units
        assert 0 <= rec < records and rec not in written
        written[int(rec)] = unit
        if i < whole or tile0 % tpu == 0:
            assert rec == unit                              # whole units and a tail unit's first tile: the unit's own record
        else:
            assert rec >= nunits
    assert (seen_tiles == 1).all() and len(written) == records
    # what rn_pair_reduce_parts reads for unit (b, jg, v): its own record, and -- walk position p = (v * jgs + jg) * B + b >= whole --
    # the records nunits + (p - whole) (tpu - 1) + t - 1, t = 1 .. tpu - 1
    per_question_extra = np.zeros(B, dtype=np.int64)
    for b in range(B):
        for jg in range(jgs):
            for v in range(nu):
                unit = (b * jgs + jg) * nu + v
                p = (v * jgs + jg) * B + b
                reads = [unit] + ([nunits + (p - whole) * (tpu - 1) + t - 1 for t in range(1, tpu)] if p >= whole else [])
                assert all(written[r] == unit for r in reads), (b, jg, v)
                per_question_extra[b] += len(reads) - 1
    assert sum(per_question_extra) == records - nunits
    if (nunits - whole) % B == 0:                           # the tail spreads evenly over the questions (question-fastest walk)
        assert per_question_extra.max() == per_question_extra.min()
    else:
        assert per_question_extra.max() - per_question_extra.min() <= tpu - 1
