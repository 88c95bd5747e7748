#!/usr/bin/env python3
"""Time the headline forward chain (f16s, factored first layer, e4m3 copies + masks + pair sums) alone at B = 64, n = 64; with
RN_DIAG=1 (diagnostics build): its timing ablations."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
import relationnetworks_clevr_amd as pkg
H = pkg.rn_hip
if os.environ.get("RN_LIB"):
    H.LIB_PATH = os.path.abspath(os.environ["RN_LIB"])
lib = H.load()
B, n, L, G, k, Q = 64, 64, 4, 256, 26, 128
M, kt = B * n * n, 2 * 26 + 128
torch.manual_seed(0)
x = torch.rand(B, n, k, device="cuda") * 2 - 1
q = torch.rand(B, Q, device="cuda") * 2 - 1
Ws = [(torch.rand(G, kt if l == 0 else G, device="cuda") - 0.5) * 0.3 for l in range(L)]
bs = [(torch.rand(G, device="cuda") - 0.5) * 0.6 for _ in range(L)]
V = H.F16S_DITHER
hi = [torch.empty(65536, dtype=torch.float16, device="cuda")] + [torch.empty(V, 65536, dtype=torch.float16, device="cuda") for _ in range(1, L)]
lo = torch.empty(65536, dtype=torch.float16, device="cuda")
w0T = torch.empty(kt, G, device="cuda")
jobs = [(Ws[0], kt, 1, G, k, hi[0], 4 | 1), (Ws[0], kt, 1, G, k, lo, 8 | 1), (Ws[0], kt, 1, G, kt, w0T, 2)]
jobs += [(Ws[l], G, 1, G, G, hi[l], 4 | (V << 8)) for l in range(1, L)]
H.pack_matrix_frag_many(jobs)
Xp = torch.empty(B * n, 64, dtype=torch.float16, device="cuda"); Vc = torch.empty(B * n, G, device="cuda")
H.pair_tables(x, q, w0T, bs[0], Xp, Vc, B, n, k, Q, G)
Hs = [torch.empty(M, G, dtype=torch.float8_e4m3fn, device="cuda") for _ in range(3)] + [None]
masks = list(torch.zeros(L, H.g_chain_rr_mask_bytes(M), dtype=torch.uint8, device="cuda"))
part = torch.empty(M // 256, G, device="cuda")


def spin_cycles(ms):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); torch.cuda._sleep(1000000); e1.record(); torch.cuda.synchronize()
    return int(1000000 * ms / max(e0.elapsed_time(e1), 1e-3))


def time_variants(variants, reps=15, rest_ms=0.8):
    """Median launch duration of every variant, measured ROUND-ROBIN with a low-power spin in front of every launch: a kernel
    launched back to back runs at the clock its own power draw leaves (this one: 215 us back to back, 166 us inside the step),
    so variants are only comparable under the same duty cycle."""
    spin = spin_cycles(rest_ms)
    ts = {k: [] for k in variants}
    for r in range(reps + 2):
        for name, (pre, fn) in variants.items():
            pre()
            torch.cuda._sleep(spin)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record()
            torch.cuda.synchronize()
            if r >= 2:
                ts[name].append(e0.elapsed_time(e1) * 1e3)
    return {k: sorted(v)[len(v) // 2] for k, v in ts.items()}


run = lambda: H.g_chain_fwd_rr_f16s_alg0(Xp, Vc, n, hi, lo, bs, Hs, masks, part, M, G)
flops = 2.0 * M * G * (kt + 3 * G)
variants = {"baseline": (lambda: None, run)}
if os.environ.get("RN_DIAG", "0") == "1":
    names = {1: "no bias rows", 2: "no barriers", 4: "no weight-stream waits", 6: "no barriers, no waits", 8: "no copy-out", 16: "no mask stores", 24: "no copy-out, no mask stores",
             64: "no weight requests", 32: "no epilogue at all (MFMAs + weight stream + fragment reads)", 96: "no epilogue, no weight requests", 88: "no copy-out, no mask stores, no weight requests",
             102: "MFMAs + fragment reads only (no epilogue, requests, barriers, waits)"}
    for abl, what in names.items():
        variants["ABL %3d (%s)" % (abl, what)] = ((lambda a=abl: lib.rn_diag_set_abl(a)), run)
    variants["baseline again"] = (lambda: lib.rn_diag_set_abl(0), run)
res = time_variants(variants)
for k_, us in res.items():
    print("%-44s %7.1f us  (%.3f of 2.5 PF algorithmic)" % (k_, us, flops / (us * 1e-6) / 2.5e15))
