#!/usr/bin/env python3
"""Headline benchmark: CLEVR questions/s of one TRAINING step (fwd + bwd + clip + Adam) of the
Relation Network `original-fp` at B=64 per GPU on the 8x8 conv grid (BASELINE.json configs[1]),
synthetic 128x128 images + random 20-token questions, random-init weights.

  python bench.py --gpus 1 --steps 20 --warmup 5
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
         bench.py --gpus N --steps K --warmup W

One JSON line on rank 0.  `value` = whole-job questions/s (all ranks, max-over-ranks time) in the
arithmetic mode the MODULE selects by itself (`precision: "auto"`; for original-fp
that is "f16s", the mode that meets the 1e-3 log-prob bar).  What the line carries besides the contract:

  parity           MEASURED in this run: the benched mode on the reference's golden fixtures (G-fp64: the
                   relational layer at the headline shape; the released checkpoint: the whole model) --
                   max-norm relative log-prob error, argmax agreement, gradient errors.  Nothing is quoted.
  roofline         the DOMINANT g_theta kernel = the one with the longest launch in the step (`kernel_key`; round 4: the
                   streaming weight gradient): ALGORITHMIC flops (2*M*sum(K_l*G); padding and second passes
                   not counted) / its launch duration from HIP events on the launch stream, against the dense 16-bit
                   MFMA peak; `traffic` (+ `hbm_frac`) from the newest profiles/*pmc_hbm_traffic.txt (`traffic_source`)
                   or null.  `gemm_chain`: the forward chain, the kernel north_star's >= 30 % target is written for
                   (`frac` algorithmic, `frac_executed` what the pipe really ran).  `kernels` / `all_g_theta`: every
                   g_theta kernel as the step runs it; `ms_serial` / `all_g_theta_serial`: with the side-stream overlap
                   off; `step`: g_theta's flops over the whole step's time.
  sustained        the same graph replayed for --sustain seconds (default 3) right after the K timed steps: rate, ms/step and the
                   rocm-smi clocks / power sampled meanwhile -- K = 20 steps are 15 ms, too short for the chip's sustained clocks.
  roofline.frac_alone / ms_alone: the dominant weight-gradient launch timed on its own (nothing beside it: what a kernel trace of
                   the eager step sees); `frac` / `ms_per_launch`: its bracket inside the step, beside the conv stack's backward.
  roofline.diagnostics.mfma_stream_probe: a bare-MFMA stream's rate on this box on patterned and on all-zero operands
                   (rn_probe_mfma_stream_ops) -- a probe of the power-limited clock, NOT a denominator: every `frac` divides by
                   the nominal 2.5 PFLOP/s; `of_probe_not_peak` values are labelled as what they are.
  convergence      (--convergence STEPS) fp32 and the benched mode trained on a learnable synthetic task with the same seeds:
                   loss curves, held-out accuracy, and the e4m3 copy guard's log.
  pair_build_k1    rn_pair_build_fwd launched on its own at the benched shape: 94.83 MB / duration vs 8 TB/s
                   (north_star's "HBM GB/s on the pair-build kernel"; the headline path itself builds two small
                   tables instead, reported as `pair_tables`).
  other_modes      the same step in the other arithmetic modes, each with its own measured parity.
  cpu_baseline     the oracle's un-fused fp32 CPU restatement of the reference model on this host (rank 0,
                   N=1 only): 3 warm-up + 5 timed fwd+bwd steps, median; `cpu_baseline_sd4`:
                   the same for BASELINE.json configs[0] (original-sd, B=4, 12-object state descriptions).
  comm             (N > 1) ranks_seen, where the gradient exchange runs (`exchange_mode`: in-graph | eager), why it fell back
                   (`exchange_fallback`), what the start-up checks measured, the all-reduce's stand-alone cost.

  python bench.py --config original-sd      # BASELINE.json configs[0] on the GPU: B=4 state descriptions (per-layer fp32 kernels)
  RN_BENCH_ONE_RANK_EXCHANGE=1 python bench.py    # diagnostic: the N > 1 code path (process group over RCCL, in-graph exchange, `comm`)
                                                  # on a ONE-rank communicator -- what a one-GPU box can execute of it"""
import argparse
import contextlib
import glob
import io
import json
import os
import re
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_TFLOPS = {"bf16": 2500.0, "f16s": 2500.0, "fp32": 157.3, "bf16x3": 2500.0}      # MI355X dense MFMA peaks (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0
DTYPE_DETAIL = {"bf16": "bf16 x bf16 MFMA, fp32 accumulate", "fp32": "fp32 MFMA (exact fmaf chains)",
                "bf16x3": "fp32 storage; g_theta forward / dgrad / weight-gradient products as hi*hi + hi*lo + lo*hi of bf16-split operands on the bf16 "
                          "MFMA pipe (fp32 accumulate, 2^-16 of a product dropped); f_phi exact fp32",
                "f16s": "fp16 activations x fp16 weights, fp32 accumulate: layer 0 hi + lo split weights (2 MFMA passes), layers 1-3 one pass on "
                        "tile-dithered weight images (4 roundings, tile t uses image t mod 4); backward bf16, e4m3 copies of H_0..2 for the weight gradients"}


class A:
    qdict_size, adict_size = 82, 28


def make_batch(B, device, hw=128, T=20, state_desc=False):
    """Synthetic batch.  From pixels: images in [0, 1) as ToTensor yields them (train.py:186).  State descriptions
    (`*-sd`): (B, 12, 7) object rows -- 3 / 6 / 10 / 12 real objects per question, cyclically, the rest zero rows: the padding of
    utils.py:101-107, which takes part in every pair."""
    g = torch.Generator().manual_seed(42)            # the reference's default seed (train.py:382)
    if state_desc:
        img = torch.randn(B, 12, 7, generator=g)
        for b, real in enumerate([(3, 6, 10, 12)[i % 4] for i in range(B)]):
            img[b, real:] = 0.0
    else:
        img = torch.rand(B, 3, hw, hw, generator=g)
    qst = torch.randint(1, 83, (B, T), generator=g)
    lab = torch.randint(0, 28, (B,), generator=g)
    return img.to(device), qst.to(device), lab.to(device)


def g_flops_fwd(M, hyp, k):
    gl, inj, Q = hyp["g_layers"], hyp["question_injection_position"], hyp["lstm_hidden"]
    tot = 0
    for l, w in enumerate(gl):
        kin = (2 * k if l == 0 else gl[l - 1]) + (Q if l == inj else 0)
        tot += 2 * M * kin * w
    return tot


def quiet_rn(pkg, hyp):
    with contextlib.redirect_stdout(io.StringIO()):
        return pkg.RN(A, hyp)


# ----------------------------------------------------------------------------------------------- checker legs
def cpu_baseline(cfg, B, hw, warm=3, steps=5):
    """The oracle's torch-CPU restatement of the reference model (same un-fused op sequence), fwd+bwd."""
    from oracle import formula, rn_oracle as O
    n_thr = min(len(os.sched_getaffinity(0)), 64)
    torch.set_num_threads(n_thr)
    torch.manual_seed(42)
    m = O.RNOracle(formula.QDICT, formula.ADICT, formula.HYP[cfg])
    m.train()
    img, qst, lab = make_batch(B, "cpu", hw, state_desc=bool(formula.HYP[cfg]["state_description"]))
    times = []
    for it in range(warm + steps):
        t0 = time.perf_counter()
        m.zero_grad()
        loss = torch.nn.functional.nll_loss(m(img, qst), lab)
        loss.backward()
        times.append(time.perf_counter() - t0)
    ts = sorted(times[warm:])
    t = ts[len(ts) // 2]
    model = "?"
    try:
        with open("/proc/cpuinfo") as f:
            model = [l.split(":", 1)[1].strip() for l in f if l.startswith("model name")][0]
    except Exception:
        pass
    return {"value": B / t, "unit": "questions/s", "cores": n_thr, "kind": "port",
            "sample": "%d timed fwd+bwd steps (median; min %.3f s, max %.3f s) of the fp32 CPU restatement at B=%d, %s, after %d warm-up steps; %s"
                      % (steps, ts[0], ts[-1], B, cfg, warm, model)}


def parity_check(pkg, cfg, prec, hw=128):
    """The checker leg: the benched arithmetic mode against the reference's golden vectors (tests/golden/*.npz, recorded
    from /root/reference/model.py by tests/golden/make_golden.py), measured NOW on this GPU.  Inputs are the fixtures'
    closed-form ones (oracle/formula.py: test-data generator, used here as the checker's input only)."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import gold
    from oracle import formula

    def l2rel(a, b):
        a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
        return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))

    out = {"mode": prec, "tolerance": 1e-3, "metric": "max|got - ref| / max|ref| on log-probs (fp32 reference, identical inputs)"}
    # the relational-layer fixture of the BENCHED shape: the 14 x 14 grid (--hw 224) has its own, recorded at its real size
    if hw == 224:
        tag_rl = "G-fp196-b32" if cfg == "original-fp" else None
    else:
        tag_rl = {"original-fp": "G-fp64", "ir-fp": "G-ir64", "original-sd": "G-sd4", "ir-sd": "G-irsd4"}.get(cfg)
    tag_ck = "pretrained_original_fp" if cfg == "original-fp" else ("pretrained_ir_fp" if cfg == "ir-fp" else None)
    worst = 0.0
    if tag_rl:
        g = gold.load(tag_rl)
        hyp, sd, x, q, lab = gold.rl_case(g["meta"])
        rl = pkg.RelationalLayer(hyp["rl_in_size"], formula.ADICT, hyp["lstm_hidden"], dict(hyp, precision=prec))
        rl.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
        rl = rl.cuda().eval()
        xt = torch.from_numpy(x).cuda().requires_grad_(True)
        qt = torch.from_numpy(q).cuda().requires_grad_(True)
        lp = rl(xt, qt)
        torch.nn.functional.nll_loss(lp, torch.from_numpy(lab).cuda()).backward()
        torch.cuda.synchronize()
        lpn = lp.detach().cpu().numpy()
        grads = {n_: p.grad.detach().cpu().numpy() for n_, p in rl.named_parameters()}
        e = gold.rel_err(lpn, g["log_probs"])
        worst = max(worst, e)
        out[tag_rl] = {"what": "relational layer, B=%d n=%d, formula weights, fwd+bwd" % (g["meta"]["b"], g["meta"]["n"]),
                       "log_prob_rel_err": e, "argmax_agree": float((lpn.argmax(1) == g["log_probs"].argmax(1)).mean()),
                       "dx_l2_rel": (l2rel(xt.grad.cpu().numpy(), g["dx"]) if "dx" in g else
                                     abs(float(np.linalg.norm(xt.grad.double().cpu().numpy())) - float(g["dx_norm"])) / float(g["dx_norm"])),
                       "dq_l2_rel": l2rel(qt.grad.cpu().numpy(), g["dq"]),
                       "bias_grads_l2_rel_max": max(l2rel(grads[k_[5:]], g[k_]) for k_ in g if k_.startswith("grad/"))}
        if "dx" in g:
            # per question: a ReLU network's gradient is discontinuous in the forward's rounding (a gate of f_phi that flips changes
            # ONE question's dx / dq by percents: DESIGN section 4.4), so the whole-batch L2 figure above is dominated by the one or
            # two such questions of the fixture; the median says what the arithmetic itself does
            pq = lambda a, r: [l2rel(a[i], r[i]) for i in range(a.shape[0])]
            ex, eq = pq(xt.grad.cpu().numpy(), g["dx"]), pq(qt.grad.cpu().numpy(), g["dq"])
            out[tag_rl]["per_question"] = {"dx_l2_rel_median": float(np.median(ex)), "dx_l2_rel_max": float(np.max(ex)),
                                           "dq_l2_rel_median": float(np.median(eq)), "dq_l2_rel_max": float(np.max(eq))}
        # the weight gradients (what the e4m3 activation copies touch): 64 sampled entries + the norm of every tensor the fixture
        # pins (gold.check_grads: max-norm relative error, worst of sample / norm)
        per = {}
        gold.check_grads(g, grads, float("inf"), per)
        out[tag_rl]["g_weight_grads_rel_err"] = {k_: per[k_] for k_ in sorted(per) if k_.startswith("g_layers") and k_.endswith("weight")}
        out[tag_rl]["activation_copies"] = "e4m3" if (prec in ("bf16", "f16s") and pkg.options.OPT.h8) else "16-bit / fp32"
    if tag_ck:
        g = gold.load(tag_ck)
        # eval_two_pass off: the checkpoint in the arithmetic the TIMED step runs (eval() would otherwise take the two-pass inference
        # arithmetic, measured beside it as `eval_default_log_prob_rel_err`)
        m = quiet_rn(pkg, dict(formula.HYP[g["meta"]["cfg"]], precision=prec, eval_two_pass=False))
        m.load_state_dict({k_[3:]: torch.from_numpy(v) for k_, v in g.items() if k_.startswith("sd/")}, strict=False)
        m.cuda(); m.eval()
        img = torch.from_numpy(formula.hash_uniform((4, 3, 128, 128), g["meta"]["img_seed"], 0.0, 1.0)).cuda()
        qst = torch.from_numpy(formula.hash_ints((4, 20), g["meta"]["qst_seed"], 1, formula.QDICT + 1)).cuda()
        with torch.no_grad():
            lpn = m(img, qst).cpu().numpy()
            m.rl.eval_two_pass = True
            lpe = m(img, qst).cpu().numpy()
        e = gold.rel_err(lpn, g["log_probs"])
        worst = max(worst, e)
        out[tag_ck] = {"what": "whole model, released checkpoint, B=4, the timed step's arithmetic", "log_prob_rel_err": e,
                       "argmax_agree": float((lpn.argmax(1) == g["log_probs"].argmax(1)).mean()),
                       "eval_default_log_prob_rel_err": gold.rel_err(lpe, g["log_probs"])}
    out["log_prob_rel_err_max"] = worst if (tag_rl or tag_ck) else None
    out["meets_1e-3"] = bool(worst <= 1e-3) if (tag_rl or tag_ck) else None
    return out


def hbm_traffic_from_profiles(kernel_pat):
    """HBM bytes per launch of the kernel whose name matches `kernel_pat`, from the newest profiles/*pmc_hbm_traffic.txt
    (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, KB; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for
    gfx950).  -> (bytes | None, file | None)"""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_hbm_traffic*.txt")))
    for f in reversed(files):
        fetch = write = None
        for line in open(f):
            if line.startswith("#") or not re.search(kernel_pat, line):
                continue
            m = re.search(r"FETCH_SIZE=([0-9.e+]+)", line)
            if m and fetch is None:
                fetch = float(m.group(1))
            m = re.search(r"WRITE_SIZE=([0-9.e+]+)", line)
            if m and write is None:
                write = float(m.group(1))
        if fetch is not None and write is not None:
            return (2.0 * fetch + write) * 1024.0, os.path.relpath(f, ROOT)
    return None, None


def time_launch(fn, n=20):
    """Duration (ms) of ONE launch from HIP events on the current stream: median bracket around TWO back-to-back launches
    minus median bracket around ONE.  A bracket around a single launch also holds the event packets' own ~5 us (and, with
    an empty queue, the host's launch latency) -- a quarter of a 20 us kernel; the difference holds one launch and agrees
    with rocprofv3's kernel durations (profiles/README.md).  A ~100 us spin kernel is queued first so that everything
    bracketed is already in the queue when the device reaches it."""
    for _ in range(3):
        fn()

    def bracket(reps):
        ts = []
        for _ in range(n):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda._sleep(200000)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        ts.sort()
        return ts[len(ts) // 2]
    return bracket(2) - bracket(1)


def sustained_mfma(H, dev):
    """DIAGNOSTIC, not a denominator: what a bare stream of v_mfma_f32_32x32x16 on every SIMD of every CU runs at on THIS device, now
    (rn_probe_mfma_stream_ops: two waves per SIMD like the chains, nothing but MFMAs, launched alone for ~0.4 ms per bracket) -- on
    patterned operands (what a real kernel multiplies) and on all-zero ones.  The chip clocks to its power budget
    (MI355X_MICROARCH.md "DVFS give-back": zero-filled inputs run +19 % TF/s on the same binary; its 2495 TF for this instruction
    is such a figure): the two rows bracket what "MFMA-bound" can mean here.  Every `frac` of this file divides by the NOMINAL dense
    peak (2.5 PFLOP/s); `of_probe` values are labelled as what they are."""
    cus = torch.cuda.get_device_properties(dev).multi_processor_count
    out = torch.empty(cus * 512, device=dev)
    res = {}
    for name, dt in (("f16", torch.float16), ("bf16", torch.bfloat16)):
        iters = 600
        for zero in (False, True):
            flops = H.probe_mfma_stream(out, cus, 2, iters, dt, zero_operands=zero)
            ms = time_launch(lambda: H.probe_mfma_stream(out, cus, 2, iters, dt, zero_operands=zero), n=7)
            res[name + ("_zero_operands" if zero else "")] = {"tflops": flops / (ms * 1e-3) / 1e12, "ns_per_mfma_slot": ms * 1e6 / (2 * iters * 16), "us_per_launch": 1e3 * ms}
    res["what"] = ("bare v_mfma_f32_32x32x16 stream, 2 waves per SIMD on every CU, launched alone (a 32-cycle slot at 2.4 GHz = 13.3 ns); "
                   "patterned operands vs all-zero operands: a probe of the power-limited clock, NOT the roofline's peak")
    return res


def wgrad_alone(H, B, n, dev, njp=None):
    """The weight-gradient launch of the chain path on its own at the benched shape (two stored gradients on e4m3 activation images
    + the gate job; random operands): HIP-event time of ONE launch (main kernel + its partial-sum reduction) with nothing beside
    it -- what a kernel trace of the eager step sees, where the in-step bracket also holds the conv stack's backward on the other
    streams."""
    njp = njp or n                                           # (the padded j axis of the chain path: rows per question = n * njp)
    G, M = 256, B * n * njp
    dZ = [((torch.rand(M, G, device=dev) - 0.5) * 1e-2).bfloat16() for _ in range(2)]
    A8 = [(torch.rand(M, G, device=dev) * 2).to(torch.float8_e4m3fn) for _ in range(3)]
    mask = torch.randint(0, 256, (H.g_chain_rr_mask_bytes(M),), dtype=torch.uint8, device=dev)
    H.relu_gate_image(mask, A8[2], M)
    dxg = torch.rand(B, G, device=dev) - 0.5
    dW = [torch.empty(G, G, device=dev) for _ in range(3)]
    db = [torch.empty(G, device=dev) for _ in range(3)]
    jobs = [(dZ[0], A8[0], dW[0], db[0]), (dZ[1], A8[1], dW[1], db[1]), (None, A8[2], dW[2], db[2])]
    return time_launch(lambda: H.g_wgrad_blocked(jobs, M, dxg=dxg, rows_per_question=n * njp), n=15)


def pair_build_k1(H, B, n, k, Q, dev):
    """K1 on its own at the benched shape: the (M, 2k+Q) bf16 pair matrix of model.py:112-127."""
    M = B * n * n
    ld = (2 * k + Q + 63) // 64 * 64
    x = torch.rand(B, n, k, device=dev)
    q = torch.rand(B, Q, device=dev)
    P = torch.empty(M, ld, dtype=torch.bfloat16, device=dev)
    ms = time_launch(lambda: H.pair_build_fwd(x, q, P, H.RN_BF16, B, n, k, Q, ld), n=60)     # (a 20 us kernel: more brackets for the two medians)
    nbytes = M * (2 * k + Q) * 2 + B * n * k * 4 + B * Q * 4             # algorithmic (un-padded) bytes
    gbs = nbytes / (ms * 1e-3) / 1e9
    return {"bound": "hbm", "kernel": "pair_build_kernel<bf16> (rn_pair.hip), launched alone", "achieved": gbs, "peak": PEAK_HBM_GBS,
            "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS, "bytes_per_launch": nbytes, "written_incl_padding": M * ld * 2,
            "us_per_launch": 1e3 * ms}


def mode_rate(pkg, dp, hyp, prec, dev, img, qst, lab, B, steps=10):
    """One more arithmetic mode: the same graph-replayed train step, timed over groups of `steps` steps (host clock, synchronised)."""
    torch.manual_seed(42)
    model = quiet_rn(pkg, dict(hyp, precision=prec))
    model.cuda(dev)
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=5e-6, weight_decay=1e-4, fused=True)
    tr = dp.DataParallelTrainer(model, opt, clip_norm=50.0, use_graph=True)
    try:
        bufs = tr.input_buffers(img, qst, lab)            # (the batch in the step graph's own tensors, like the headline line)
        for d_, s_ in zip(bufs, (img, qst, lab)):
            if d_ is not s_:
                d_.copy_(s_)
        img, qst, lab = bufs
        for _ in range(3):
            tr.step(img, qst, lab)
    except RuntimeError as e:
        return {"unsupported": str(e)[:120]}
    dts = []
    for _ in range(3):                                   # three groups of `steps`, the median one (a one-off host stall in a 10-step
        torch.cuda.synchronize()                         # window once read 10x slow)
        t0 = time.perf_counter()
        for _ in range(steps):
            tr.step(img, qst, lab)
        torch.cuda.synchronize()
        dts.append(time.perf_counter() - t0)
    dt = sorted(dts)[1]
    return {"value": B * steps / dt, "unit": "questions/s", "ms_per_step": 1e3 * dt / steps}


class ClockSampler:
    """rocm-smi's sclk / mclk / power of device 0, sampled every ~0.25 s by a thread while the sustained run is going."""

    def __init__(self, period=0.25):
        import threading
        self.period, self.samples, self._stop = period, [], threading.Event()
        self._thr = threading.Thread(target=self._run, daemon=True)

    def _run(self):
        import subprocess
        while not self._stop.is_set():
            try:
                r = subprocess.run(["rocm-smi", "-d", "0", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=5)
                d = json.loads(r.stdout)
                c = next(iter(d.values()))
                row = {}
                for k_, v in c.items():
                    m = re.search(r"\((\d+)Mhz\)", str(v))
                    if "sclk" in k_ and m:
                        row["sclk_mhz"] = int(m.group(1))
                    elif "mclk" in k_ and m:
                        row["mclk_mhz"] = int(m.group(1))
                    elif "Power" in k_:
                        try:
                            row["power_w"] = float(v)
                        except (TypeError, ValueError):
                            pass
                if row:
                    self.samples.append((time.perf_counter(), row))
            except Exception:
                pass
            self._stop.wait(self.period)

    def start(self):
        self._thr.start()

    def stop(self):
        self._stop.set()
        self._thr.join(timeout=10)

    def summary(self):
        if not self.samples:
            return None
        out = {"samples": len(self.samples)}
        for key in ("sclk_mhz", "mclk_mhz", "power_w"):
            vals = [r[key] for _t, r in self.samples if key in r]
            if vals:
                out[key] = {"min": min(vals), "max": max(vals), "mean": sum(vals) / len(vals), "last": vals[-1]}
        return out


def timed_steps(step, steps, warmup, sync):
    """W untimed warm-up steps, then EXACTLY K steps bracketed by sync() on both sides -> (seconds, last step's result).
    Shared by the real run and the launch-plumbing dry run."""
    last = None
    for _ in range(warmup):
        last = step()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        last = step()
    sync()
    return time.perf_counter() - t0, last


def max_over_ranks(dt, world, device):
    """The job's time is the slowest rank's."""
    import torch.distributed as dist
    if world > 1 or (dist.is_available() and dist.is_initialized()):
        tt = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    return dt


# ---- a failed run still prints ONE JSON line on rank 0 (the first real N > 1 run is a one-shot -- whatever goes
# wrong, the line says how far the job got: ranks seen, where the gradient exchange was running, why it fell back)
FAIL = {"rank": 0, "world": 1, "args": None, "trainer": None, "ranks_seen": None, "phase": "start-up", "printed": False, "backend": None}


def failure_line(reason):
    a, tr = FAIL["args"], FAIL["trainer"]
    comm = {"ranks_seen": FAIL["ranks_seen"], "exchange_mode": None, "exchange_fallback": None, "exchange_checks": None,
            "allreduce_us_per_step": None, "backend": FAIL["backend"]}
    if tr is not None:
        try:
            comm.update({"exchange_mode": tr.exchange_mode(), "exchange_fallback": tr.exchange_fallback, "exchange_checks": tr.exchange_checks})
        except Exception:
            pass
    return {"metric": "CLEVR questions/sec (train fwd+bwd)", "value": None, "unit": "questions/s", "n_gpus": FAIL["world"],
            "steps": getattr(a, "steps", None), "warmup": getattr(a, "warmup", None), "ms_per_step": None, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "data": "synthetic", "failed": True, "error": str(reason)[:600], "phase": FAIL["phase"],
            "comm": comm}


def emit_line(obj):
    """The ONE JSON line, as the LAST line of stdout: C-level stdio is flushed first (RCCL prints a version banner through printf
    when a communicator is created; buffered, it would otherwise land behind this line when the process exits)."""
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.write(json.dumps(obj) + "\n")
    sys.stdout.flush()


def print_failure(reason):
    if FAIL["rank"] == 0 and not FAIL["printed"]:
        FAIL["printed"] = True
        try:
            emit_line(failure_line(reason))
        except Exception:
            pass


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch (default: 64; 4 for the state-description configs, BASELINE.json configs[0])")
    ap.add_argument("--config", default="original-fp")
    ap.add_argument("--precision", default=os.environ.get("RN_PRECISION", "auto"), choices=["auto", "bf16", "f16s", "fp32", "bf16x3"],
                    help='"auto" (default) = what the module selects by itself for this config')
    ap.add_argument("--hw", type=int, default=128, help="image side (224 -> 14x14 grid stress config)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-other-modes", action="store_true", help="skip the extra timing of the other arithmetic modes")
    ap.add_argument("--no-parity", action="store_true", help="skip the live parity check against the golden fixtures")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel eagerly instead of replaying a hipGraph")
    ap.add_argument("--convergence", type=int, default=0, metavar="STEPS",
                    help="also train STEPS steps on the learnable synthetic task in fp32 and in the benched mode (same seeds) and report "
                         "both loss curves + held-out accuracy as `convergence` (train.convergence_run; ~15 s per mode at 300 steps)")
    ap.add_argument("--sustain", type=float, default=3.0,
                    help="seconds of back-to-back steps AFTER the timed K steps, same graph: reported as `sustained` (0: skip)")
    args = ap.parse_args()

    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("RN_BENCH_BACKEND", "nccl")     # "gloo" + RN_BENCH_DRY=1: launch-plumbing dry run on CPU (tests)
    dry = os.environ.get("RN_BENCH_DRY", "0") == "1"
    FAIL.update(rank=rank, world=world, args=args, backend=backend)
    if world > 1:
        import signal

        def _terminated(signum, frame):                 # (torch.distributed.run tears the job down when ANOTHER rank died)
            print_failure("terminated by signal %d in phase '%s' (another rank failed?)" % (signum, FAIL["phase"]))
            os._exit(128 + signum)
        signal.signal(signal.SIGTERM, _terminated)
    if args.gpus > 1 and world == 1:
        raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d ..." % (args.gpus, args.gpus))
    if dry:
        args.batch = args.batch or 64
        return dry_run(args, world, rank, backend)
    if os.environ.get("RN_BENCH_SAME_GPU", "0") == "1":     # tests: every rank on device 0 (with RN_BENCH_BACKEND=gloo: RCCL refuses that)
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # RN_BENCH_ONE_RANK_EXCHANGE=1 (diagnostic, N = 1 only): a ONE-rank communicator and DataParallelTrainer(single_rank_exchange=True)
    # -- every line of the N > 1 path below (roll call, barriers, the in-graph exchange with its self-check, `comm`) runs over RCCL on
    # a one-GPU box; the sum over one rank is the identity, so `value` must equal the plain line's
    one_rank = world == 1 and os.environ.get("RN_BENCH_ONE_RANK_EXCHANGE", "0") == "1"
    multi = world > 1 or one_rank
    if multi:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if one_rank:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29517")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group(backend, device_id=dev)
        else:
            dist.init_process_group(backend)

    import relationnetworks_clevr_amd as pkg
    from relationnetworks_clevr_amd import dp
    H = pkg.rn_hip
    H.load()
    hyps = json.load(open(os.path.join(ROOT, "relationnetworks-clevr_amd", "config.json")))["hyperparams"]
    base_hyp = dict(hyps[args.config])
    state_desc = bool(base_hyp["state_description"])
    B = args.batch if args.batch is not None else (4 if state_desc else 64)      # (BASELINE.json configs[0]: original-sd at B = 4)
    d = args.hw // 16
    n, k = (d * d, base_hyp["rl_in_size"] // 2) if not state_desc else (12, base_hyp["rl_in_size"] // 2)
    M = B * n * n
    torch.manual_seed(42)
    model = quiet_rn(pkg, dict(base_hyp, precision=args.precision))
    prec = model.rl.resolved_precision(B, n, k)              # what "auto" turns into for this shape
    hyp = dict(base_hyp, precision=prec)
    model.cuda(dev)
    model.train()
    try:
        opt = torch.optim.Adam(model.parameters(), lr=5e-6, weight_decay=1e-4, fused=True)   # train.py:330,376
    except Exception:
        opt = torch.optim.Adam(model.parameters(), lr=5e-6, weight_decay=1e-4, foreach=True)
    use_graph = not args.no_graph
    # copy_guard_every=0: the e4m3 copy guard (an eager forward + a host sync every 512 steps) runs ONCE, in front of the capture, and
    # then stays out of the timed / sustained windows; train.py keeps the periodic guard
    trainer = dp.DataParallelTrainer(model, opt, clip_norm=50.0, use_graph=use_graph, copy_guard_every=0, single_rank_exchange=one_rank)
    FAIL.update(trainer=trainer, phase="trainer built")
    if multi:
        dp.Watchdog.on_expire = lambda what, seconds: print_failure("watchdog: no progress for %.0f s in: %s" % (seconds, what))
    img, qst, lab = make_batch(B, dev, args.hw, state_desc=state_desc)
    if use_graph:
        trainer.check_activation_copies(img, qst, lab)
    wd, wd_s = trainer.watchdog, (float(pkg.options.OPT.dp_timeout) if multi else 0.0)
    ranks_seen = None
    if multi:
        with wd.guard("bench: roll call of the ranks", wd_s):
            ranks_seen = trainer.ctl.ranks_seen()
        FAIL.update(ranks_seen=ranks_seen, phase="roll call done")
        if ranks_seen != list(range(world)):
            raise SystemExit("bench: expected ranks %r, saw %r" % (list(range(world)), ranks_seen))

    def sync():
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
            torch.cuda.synchronize()

    # The batch is resident in HBM when the timed region starts -- in the step graph's OWN input tensors (trainer.input_buffers: where
    # train.py's loader lands its host -> device copies), so a step is the replay alone.  A caller that brings every batch in tensors
    # of its own pays one more launch per step (rn_copy_many, ~14 us): that rate is measured too and reported as `with_batch_copy`.
    img0, qst0, lab0 = img, qst, lab
    if use_graph:
        with wd.guard("bench: capture of the step graph", 2.0 * wd_s):
            bufs = trainer.input_buffers(img, qst, lab)
        for d_, s_ in zip(bufs, (img, qst, lab)):
            if d_ is not s_:
                d_.copy_(s_)
        img, qst, lab = bufs
    # ---- timed region: W warm-up steps, then exactly K steps, barrier + synchronize on both sides
    FAIL["phase"] = "step graph captured (exchange: %s)" % trainer.exchange_mode()
    H.TIMER.enabled = False
    with wd.guard("bench: warm-up + timed region (%d + %d steps)" % (args.warmup, args.steps), wd_s):
        dt, loss = timed_steps(lambda: trainer.step(img, qst, lab), args.steps, args.warmup, sync)
    with_copy = None
    if use_graph and img0 is not img:
        # (both forms once more, back to back: the K timed steps above start 5 replays after an idle chip and run a few percent under
        # the rate the chip reaches some milliseconds later -- the pair below is measured in ONE clock state)
        dt_a, _ = timed_steps(lambda: trainer.step(img, qst, lab), args.steps, 2, sync)
        dt_c, _ = timed_steps(lambda: trainer.step(img0, qst0, lab0), args.steps, 2, sync)
        dt_a, dt_c = max_over_ranks(dt_a, world, dev), max_over_ranks(dt_c, world, dev)
        with_copy = {"value": world * B * args.steps / dt_c, "ms_per_step": 1e3 * dt_c / args.steps,
                     "same_moment_without_copy": world * B * args.steps / dt_a,
                     "what": "K more steps with the batch in tensors of the caller's own (one rn_copy_many launch in front of every replay), "
                             "right behind K more steps without that copy: compare these two, not with `value`"}
    sustained = None
    if args.sustain > 0:
        # The contract's K = 20 steps last ~15 ms -- too short for the chip to reach its sustained clocks under matrix load.
        # The same replay for `--sustain` seconds (step count fixed beforehand from the timed region's rate, so that every
        # rank runs the same number), bracketed like the timed region; rocm-smi clocks sampled by a thread meanwhile (rank 0).
        n_sus = max(int(args.sustain / max(max_over_ranks(dt, world, dev) / args.steps, 1e-6)), args.steps)   # (the SAME count on every rank)
        clocks = ClockSampler() if rank == 0 else None
        if clocks:
            clocks.start()
        with wd.guard("bench: sustained run (%d steps)" % n_sus, wd_s + 2.0 * args.sustain if wd_s else 0.0):
            dt_s, _ = timed_steps(lambda: trainer.step(img, qst, lab), n_sus, 0, sync)
        if clocks:
            clocks.stop()
        dt_s = max_over_ranks(dt_s, world, dev)
        sustained = {"value": world * B * n_sus / dt_s, "unit": "questions/s", "ms_per_step": 1e3 * dt_s / n_sus, "steps": n_sus,
                     "seconds": dt_s, "clocks": clocks.summary() if clocks else None}
    # the timed and sustained windows ran with the trainer's periodic guard off: validate them now -- a hand-off inside
    # the feature-split f_phi launch that was not answered leaves an error word, and garbage results since then
    fphi_status = H.f_phi_split_status(dev) if torch.cuda.is_available() else 0
    if multi:
        fphi_status = int(max_over_ranks(float(fphi_status), world, dev))
    if fphi_status:
        raise RuntimeError("rn_f_phi_split: a hand-off inside the launch was not answered during the timed / sustained window "
                           "(stage %d): the measured steps are invalid" % (fphi_status - 1))
    comm = None
    if multi:
        # attribution (outside the timed region): K more steps with event brackets around the gradient all-reduce and the fused
        # 1/world + clip + Adam -- the only parts of a step that exist because of the other ranks
        if getattr(trainer, "_opt_in_graph", False):
            # the exchange is a node of the replayed graph (events cannot bracket it there): the same all-reduce on a scratch
            # bucket of the same size, eagerly, back to back with a barrier in front -- its stand-alone cost
            scratch = torch.zeros_like(trainer.bucket.flat)
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
            dist.all_reduce(scratch)
            sync()
            for e0_, e1_ in ev:
                e0_.record(); dist.all_reduce(scratch); e1_.record()
            torch.cuda.synchronize()
            ar = sorted(e0_.elapsed_time(e1_) for e0_, e1_ in ev)
            comm = {"allreduce_us_per_step": 1e3 * ar[len(ar) // 2], "optimizer_us_per_step": None, "allreduce_bytes": 4 * trainer.bucket.numel,
                    "exchange": "inside the replayed step graph (all-reduce + 1/world + clip + Adam); the figure is the same collective launched alone"}
            comm["allreduce_us_per_step"] = max_over_ranks(comm["allreduce_us_per_step"], world, dev)
        else:
            trainer.timing = []
            for _ in range(args.steps):
                trainer.step(img, qst, lab)
            torch.cuda.synchronize()
            ar = sorted(e[0].elapsed_time(e[1]) for e in trainer.timing)
            op = sorted(e[1].elapsed_time(e[2]) for e in trainer.timing)
            trainer.timing = None
            comm = {"allreduce_us_per_step": 1e3 * ar[len(ar) // 2], "optimizer_us_per_step": 1e3 * op[len(op) // 2],
                    "allreduce_bytes": 4 * trainer.bucket.numel, "exchange": "eager, behind the replayed fwd + bwd graph"}
            for k_ in ("allreduce_us_per_step", "optimizer_us_per_step"):      # the slowest rank's
                comm[k_] = max_over_ranks(comm[k_], world, dev)
    if comm is not None:
        comm.update({"ranks_seen": ranks_seen, "exchange_mode": trainer.exchange_mode(), "exchange_fallback": trainer.exchange_fallback,
                     "exchange_checks": trainer.exchange_checks, "backend": backend, "watchdog_s": wd_s})
    ksum_step = ksum = None
    if not args.no_kernel_timing:
        # HIP events cannot bracket kernels inside a graph replay: the same K steps are repeated eagerly (same kernels, same
        # shapes, same streams) with event brackets on the launch stream -- once as the step runs them (wgrads on the side
        # stream beside the pair reduction: `in_step`), once with that overlap off so that every kernel's duration is its own
        trainer.use_graph = False
        overlap_default = pkg.options.OPT.wgrad_overlap
        # Eager launches are host-bound (~110 launches per step): with an idle GPU a bracket would also measure the host's
        # launch latency.  A one-thread spin kernel (torch.cuda._sleep) in front of every step keeps the GPU ~10 ms behind the
        # host (a slow host needs > 3 ms for the ~110 launches and their event packets: brackets read 10 % long on such a box),
        # so the whole step is queued before it starts and a bracket spans GPU time only.
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); torch.cuda._sleep(1000000); e1.record(); torch.cuda.synchronize()
        spin = int(1000000 * 10.0 / max(e0.elapsed_time(e1), 1e-3))
        for tag in ("in_step", "alone"):
            if tag == "alone":
                pkg.options.OPT.wgrad_overlap = False
            trainer.step(img, qst, lab)
            H.TIMER.enabled = True
            H.TIMER.reset()
            for _ in range(args.steps):
                torch.cuda._sleep(spin)
                trainer.step(img, qst, lab)
            H.TIMER.enabled = False
            if tag == "in_step":
                ksum_step = H.TIMER.summary(steps=args.steps)
            else:
                ksum = H.TIMER.summary(steps=args.steps)
                pkg.options.OPT.wgrad_overlap = overlap_default
        trainer.use_graph = use_graph
    # every rank empties its C-level and Python stdout buffers HERE, in front of a collective: whatever a library printed on any rank
    # (RCCL's banner) is out before rank 0 emits the one JSON line behind that collective
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()
    dt = max_over_ranks(dt, world, dev)
    if not torch.isfinite(loss).item():
        raise SystemExit("non-finite loss")

    if rank == 0:
        fwd = g_flops_fwd(M, hyp, k)
        out = {
            "metric": "CLEVR questions/sec (train fwd+bwd) at B=64, 8x8 grid" if (B == 64 and n == 64) else
                      "CLEVR questions/sec (train fwd+bwd) at B=%d, n=%d objects" % (B, n),
            "value_definition": "K hipGraph replays with the batch resident in the step graph's own input tensors (since round 4; rounds 1-3 "
                                "included one rn_copy_many hand-off launch per step: that form is `with_batch_copy`); the e4m3 copy guard ran "
                                "once before the capture and is not in any timed window",
            "value": world * B * args.steps / dt, "unit": "questions/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": prec, "dtype_detail": DTYPE_DETAIL[prec], "data": "synthetic",
            "config": {"workload": ("%s train step (LSTM+RN fwd/bwd, clip 50, Adam), B=%d/GPU, 12-object state descriptions (n=12, M=%d pair rows/GPU, "
                                    "3/6/10/12 real objects + zero rows), 20-token questions, random-init weights" % (args.config, B, M)) if state_desc else
                                   ("%s train step (conv+LSTM+RN fwd/bwd, clip 50, Adam), B=%d/GPU, %dx%d grid (n=%d, M=%d pair rows/GPU), "
                                    "synthetic %dx%d images + 20-token questions, random-init weights"
                                    % (args.config, B, d, d, n, M, args.hw, args.hw)),
                       "global_batch": world * B, "parallelism": "dp%d" % world,
                       "precision_requested": args.precision, "precision_resolved": prec,
                       "wgrad_activation_copies": ("e4m3 (H_0..2 kept for dW_1..3 only; RN_H8=0: 16-bit)"
                                                   if prec in ("bf16", "f16s") and pkg.options.OPT.h8 else "as the mode's storage type"),
                       "options_non_default": pkg.options.OPT.non_default(),
                       "launch": ("eager" if not use_graph else
                                  "one hipGraph replay per step: fwd + bwd + all-reduce + clip + Adam" if (getattr(trainer, "_opt_in_graph", False) and multi) else
                                  "one hipGraph replay per step: fwd + bwd + clip + Adam" if getattr(trainer, "_opt_in_graph", False) else
                                  "hipGraph replay of fwd+bwd, eager all-reduce/clip/Adam")},
            "loss": float(loss.detach()),
        }
        if args.convergence > 0 and world == 1:
            from relationnetworks_clevr_amd import train as T
            out["convergence"] = {"task": "train.SyntheticRelationalTask (colour / quadrant of one square; 25-step mean losses), B=64, Adam lr 1e-3, clip 50",
                                  "fp32": T.convergence_run("fp32", steps=args.convergence, model_name=args.config),
                                  prec: T.convergence_run(args.precision, steps=args.convergence, model_name=args.config)}
        out["config"]["batch_hand_off"] = ("the batch sits in the step graph's own input tensors (DataParallelTrainer.input_buffers): a step = the replay"
                                          if with_copy else "the step receives the caller's tensors")
        if with_copy:
            out["with_batch_copy"] = with_copy
        if sustained:
            sustained["vs_value"] = sustained["value"] / out["value"]
            out["sustained"] = sustained
        if comm:
            out.update({k_: comm[k_] for k_ in ("allreduce_us_per_step", "optimizer_us_per_step", "allreduce_bytes")})
            out["comm"] = comm
        if world == 1 and not args.no_parity:
            out["parity"] = parity_check(pkg, args.config, prec, args.hw)
        if ksum:
            names = ("g_fwd", "g_dgrad", "g_wgrad")
            # primary: the brackets of the step as it runs (`in_step`: the launch durations over the timed region's own launch
            # sequence; the wgrad brackets include their contention with the pair reduction).  Secondary (`ms_serial`): the same
            # step with the side-stream overlap off.  Launching one kernel many times back to back instead is NOT comparable:
            # under sustained matrix load the chip clocks down (forward chain 246 us back to back vs 196 us inside the step).
            per = {kk: (ksum_step[kk][1] / args.steps) for kk in names if kk in ksum_step}
            per_step = {kk: (ksum[kk][1] / args.steps) for kk in names if kk in ksum}
            g_launch = sum(ksum.get(kk, (0, 0.0))[0] for kk in names) // args.steps
            peak = PEAK_TFLOPS[prec]
            inj_l = hyp["question_injection_position"]
            # ONE shape predicate, the module's own (functional.chain_ok): the register-resident chains run iff the mode is "f16s"
            RFm = pkg.functional
            plan = model.rl._plan(k)
            alg0 = prec == "f16s" and RFm.chain_ok(plan, B, n)
            # flops of each entry = the ALGORITHMIC flops of exactly what its launches compute (2 M K_l G per layer and
            # direction).  Chain path: the `g_wgrad` launch is dW_1..L-1 only -- layer 0's weight gradient is another launch on
            # another stream (`wgrad0_from_reductions`, its own entry below); the per-layer path's g_wgrad bracket holds every layer.
            fl0 = g_flops_fwd(M, dict(hyp, g_layers=hyp["g_layers"][:1]), k)          # layer 0: 2 M K_0 G
            fl = {"g_fwd": fwd, "g_dgrad": float(fwd - fl0), "g_wgrad": float(fwd - fl0) if alg0 else float(fwd)}
            kern = {kk: {"algorithmic_flops": fl[kk], "ms": per[kk], "achieved_tflops": fl[kk] / (per[kk] * 1e-3) / 1e12,
                         "frac": fl[kk] / (per[kk] * 1e-3) / 1e12 / peak, "ms_serial": per_step.get(kk)} for kk in per}
            w0 = ksum_step.get("g_wgrad0")                         # (rn_hip.wgrad0_from_reductions has a timer key of its own)
            if alg0 and w0 and w0[0] > 0 and w0[1] > 0:
                ms0 = w0[1] / args.steps
                kern["g_wgrad0"] = {"algorithmic_flops": float(fl0), "ms": ms0, "achieved_tflops": None,
                                    "frac": None, "traffic": None, "traffic_source": None, "ms_serial": (ksum.get("g_wgrad0", (0, 0.0))[1] / args.steps) or None,
                                    "what": "dW_0, db_0 = [Rj^T X | Ri^T X | Rq^T q] from the backward chain's pair reductions (wgrad0_part / finish kernels, "
                                            "rn_pair.hip): the factored first layer EXECUTES ~0.1 GFLOP for these algorithmic flops -- latency-bound: no rate, no roofline fraction is quoted for it"}
            njp = (n if inj_l else RFm.padded_j(n)) if alg0 else n
            # executed flops: the factored first layer runs K = 64 on chip (two passes: hi + lo weights) and the question, wherever
            # it is injected, enters as a bias row, so layers 1..3 are K = 256 products (one pass on tile-dithered images).
            # The algorithmic count stays the reference formulation's (model.py:130-152).
            Mx = B * n * njp if alg0 else M                           # (executed rows: the padded pair space where n % 32 != 0)
            executed = 2.0 * Mx * 256 * (2 * 64 + 3 * 256) if alg0 else float(fwd)
            kname = "g_chain_rr_f16s_kernel" if alg0 else None
            # per-kernel HBM traffic of the profiled shape (original-fp, B=64, 8x8) from the newest profiles/*pmc_hbm_traffic.txt
            pats = {"g_fwd": r"g_chain_rr_f16s_kernel<4, true", "g_dgrad": r"g_chain_rr_bwd_kernel<", "g_wgrad": r"wgrad_blocked_kernel<"}
            # algorithmic HBM bytes of the chain path's kernels (what each must move with the layout as it is)
            Gw = 256
            alg_bytes = {"g_fwd": Mx * Gw * 3 + 4 * Mx * 32, "g_dgrad": Mx * Gw * 2 * 2 + 4 * Mx * 32,
                         "g_wgrad": Mx * Gw * (3 + 3 + 1)} if alg0 else {}
            for kk in per:
                tr_, src_ = (None, None)
                if alg0 and B == 64 and n == 64 and inj_l == 0:
                    tr_, src_ = hbm_traffic_from_profiles(pats[kk])
                kern[kk]["traffic"], kern[kk]["traffic_source"] = tr_, src_
                kern[kk]["hbm_frac"] = (tr_ / (per[kk] * 1e-3) / 1e9 / PEAK_HBM_GBS) if tr_ else None
                if kk in alg_bytes:
                    kern[kk]["algorithmic_hbm_bytes"] = alg_bytes[kk]
            # DIAGNOSTIC: what a bare MFMA stream runs at on this box right now (the chip clocks to its power budget) -- a probe,
            # not a peak: nothing below divides `frac` by it
            sus = sustained_mfma(H, dev) if prec == "f16s" else None
            if sus:
                for kk in per:
                    kern[kk]["of_probe_not_peak"] = kern[kk]["achieved_tflops"] / sus["f16" if kk == "g_fwd" else "bf16"]["tflops"]
            # the dominant weight-gradient launch ALONE (nothing beside it): what a kernel trace of the eager step reports for it
            if alg0 and "g_wgrad" in kern and not inj_l and world == 1:
                ms_a = wgrad_alone(H, B, n, dev, njp)
                kern["g_wgrad"]["ms_alone"] = ms_a
                kern["g_wgrad"]["frac_alone"] = fl["g_wgrad"] / (ms_a * 1e-3) / 1e12 / peak
            ach_ex = executed / (per["g_fwd"] * 1e-3) / 1e12
            # all of g_theta = the three big kernels + the launches that execute layer 0's share of the algorithmic flops on the
            # chain path (tables of the factored first layer, partial-sum add-up + dx / dq, dW_0 from the reductions)
            small = ("pair_build", "pair_reduce", "g_wgrad0") if alg0 else ()
            g_ms = sum(per.values()) + sum(ksum_step[kk][1] / args.steps for kk in small if kk in ksum_step)
            g_ms_step = sum(per_step.values()) + sum(ksum[kk][1] / args.steps for kk in small if kk in ksum)
            knames = {"g_fwd": ("%s<ALG0> (rn_chain_rr.hip): 4-layer g_theta forward chain + pair sum, 1 launch/step" % kname) if kname else
                               "g_theta forward kernels (%s per-layer path, rn_gemm.hip)" % prec,
                      "g_dgrad": "g_chain_rr_bwd_kernel<RED> (rn_chain_rr.hip): backward chain (3 dgrads, ReLU gates, pair-axis reductions), 1 launch/step"
                                 if alg0 else "g_theta dgrad kernels (%s per-layer path)" % prec,
                      "g_wgrad": "wgrad_blocked_kernel (rn_wgrad_blocked.hip): dW_1..3 + db_1..3 (NOT layer 0: kernels.g_wgrad0) in one launch (+ its partial-sum reduction)"
                                 if alg0 else "g_theta wgrad kernels (%s per-layer path, rn_wgrad.hip)" % prec}
            # THE roofline of this line = the g_theta kernel that takes the most time in the step (not the one that
            # scores best); the forward chain -- the GEMM chain north_star's >= 30 % target is written for -- is `gemm_chain` beside it
            dom = max(per, key=lambda kk: per[kk])
            kd = kern[dom]
            chain = dict(kern["g_fwd"], kernel=knames["g_fwd"], frac_executed=ach_ex / peak, achieved_executed=ach_ex,
                         executed_flops_per_launch=executed)
            if sus:
                chain["executed_of_probe_not_peak"] = ach_ex / sus["f16"]["tflops"]
            out["roofline"] = {"bound": "mfma", "achieved": kd["achieved_tflops"], "peak": peak, "unit": "TFLOP/s", "frac": kd["frac"],
                               "kernel": knames[dom], "kernel_key": dom,
                               "why_this_kernel": "longest g_theta kernel of the step (%.1f us of %.1f us of g_theta kernels)" % (1e3 * per[dom], 1e3 * g_ms),
                               "algorithmic_flops_per_launch": fl[dom], "ms_per_launch": per[dom],
                               "traffic": kd["traffic"], "traffic_source": kd["traffic_source"], "hbm_frac": kd["hbm_frac"],
                               "algorithmic_hbm_bytes": kd.get("algorithmic_hbm_bytes"),
                               "ms_alone": kd.get("ms_alone"), "frac_alone": kd.get("frac_alone"),
                               "diagnostics": {"mfma_stream_probe": sus, "dominant_kernel_of_probe_not_peak": kd.get("of_probe_not_peak")},
                               "gemm_chain": chain,
                               "kernels": kern,
                               "all_g_theta": {"algorithmic_flops_per_step": 3 * fwd, "ms_per_step": g_ms, "launches_per_step": g_launch,
                                               "achieved": 3 * fwd / (g_ms * 1e-3) / 1e12, "frac": 3 * fwd / (g_ms * 1e-3) / 1e12 / peak,
                                               "timing": "as the step runs them: wgrads on a side stream beside the pair reduction"},
                               "all_g_theta_serial": {"ms_per_step": g_ms_step, "frac": 3 * fwd / (g_ms_step * 1e-3) / 1e12 / peak,
                                                      "timing": "side-stream overlap off (every kernel alone on the chip; eager launches)"},
                               "step": {"algorithmic_flops": 3 * fwd, "ms": 1e3 * dt / args.steps,
                                        "frac": 3 * fwd / (dt / args.steps) / 1e12 / peak,
                                        "what": "g_theta's algorithmic flops over the WHOLE step's time (conv / LSTM / optimiser included)"},
                               "breakdown_ms_per_step": {kk: v[1] / args.steps for kk, v in sorted(ksum_step.items())},
                               "breakdown_ms_per_step_serial": {kk: v[1] / args.steps for kk, v in sorted(ksum.items())}}
            ach = kern["g_fwd"]["achieved_tflops"]
            out["frac_algorithmic"], out["frac_executed"] = ach / peak, ach_ex / peak
            pb = ksum_step.get("pair_build")
            if pb and pb[0] > 0 and pb[1] > 0:
                esz = 4 if prec in ("fp32", "bf16x3") else 2
                Q = hyp["lstm_hidden"] if hyp["question_injection_position"] == 0 else 0
                nbytes = M * (2 * k + Q) * esz + B * n * k * 4 + B * Q * 4
                key = "pair_build"
                if alg0:                                   # rn_pair_tables instead of the pair matrix: object rows + bias rows
                    nbytes = B * n * (64 * 2 + hyp["g_layers"][0] * 4) + B * n * k * 4 + B * Q * 4 + (2 * k + Q) * hyp["g_layers"][0] * 4
                    key = "pair_tables"
                    pb = (pb[0], pb[1])
                gbs = nbytes / (pb[1] / pb[0] * 1e-3) / 1e9
                out[key] = {"bound": "hbm (latency-bound at this size)", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                            "frac": gbs / PEAK_HBM_GBS, "bytes_per_launch": nbytes, "us_per_launch": 1e3 * pb[1] / pb[0]}
            if not base_hyp["state_description"]:
                out["pair_build_k1"] = pair_build_k1(H, B, n, k, hyp["lstm_hidden"] if hyp["question_injection_position"] == 0 else 0, dev)
        if world == 1 and not args.no_other_modes:
            others = {}
            for p2 in ("f16s", "fp32", "bf16"):                         # ("bf16": the per-layer kernels -- slower AND less accurate than the default)
                if p2 == prec:
                    continue
                r = mode_rate(pkg, dp, base_hyp, p2, dev, img, qst, lab, B)
                if "unsupported" not in r and not args.no_parity:
                    r["parity"] = parity_check(pkg, args.config, p2, args.hw)
                others[p2] = r
            if prec in ("bf16", "f16s") and pkg.options.OPT.h8:
                # the benched mode with 16-bit instead of e4m3 copies of H_0..2 (and a stored last-layer gradient): what the e4m3
                # copies buy, and what they cost in weight-gradient error (parity block of both)
                with pkg.options.override(h8=False):
                    r = mode_rate(pkg, dp, base_hyp, prec, dev, img, qst, lab, B)
                    if not args.no_parity:
                        r["parity"] = parity_check(pkg, args.config, prec, args.hw)
                others[prec + ", 16-bit activation copies"] = r
            out["other_modes"] = others
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.config, B, args.hw)
            if not (args.config == "original-sd" and B == 4):
                # BASELINE.md section 3's second CPU figure: configs[0], the reference's own CPU-runnable case (~10 ms per step)
                out["cpu_baseline_sd4"] = cpu_baseline("original-sd", 4, 128, warm=5, steps=50)
        emit_line(out)
        FAIL["printed"] = True
    if multi:
        dist.destroy_process_group()


def dry_run(args, world, rank, backend):
    """RN_BENCH_DRY=1: the launch plumbing of the N > 1 bench -- env parsing, process-group init, barrier-bracketed timed
    region, MAX-over-ranks reduction, one JSON line on rank 0 -- on CPU tensors over `gloo`, with the train step replaced by
    a sleep.  Exists so that tests can execute this file under torch.distributed.run without GPUs; never a measurement."""
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group(backend)

    def sync():
        if world > 1:
            dist.barrier()

    FAIL["phase"] = "dry run: process group up"
    if os.environ.get("RN_BENCH_DRY_FAIL_RANK", "") == str(rank):      # (tests: one rank dies -- rank 0 must still print its failure line)
        raise RuntimeError("injected failure on rank %d" % rank)
    # ranks take different times per step: the MAX over ranks must be what is reported
    dt, _ = timed_steps(lambda: time.sleep(0.001 * (rank + 1)), args.steps, args.warmup, sync)
    dt = max_over_ranks(dt, world, "cpu")
    if rank == 0:
        print(json.dumps({"metric": "DRY RUN (launch plumbing only)", "value": world * args.batch * args.steps / dt, "unit": "questions/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "none", "data": "none",
                          "config": {"workload": "dry run", "global_batch": world * args.batch, "parallelism": "dp%d" % world}}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    try:
        main()
    except SystemExit as e:
        if e.code not in (0, None):
            print_failure("exit: %s" % (e.code,))
        raise
    except BaseException as e:
        print_failure("%s: %s" % (type(e).__name__, e))
        raise
