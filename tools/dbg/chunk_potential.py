#!/usr/bin/env python3
"""Large pair matrices (B = 640 at n = 64; the 14 x 14 grid at B = 32): dZ_1, dZ_2 no longer fit the 256-MB Infinity Cache between
the backward chain that writes them and the weight gradient that reads them.  What would a CHUNKED order buy -- backward chain and
weight gradient back to back per chunk of C questions, the chunk's dZ in a buffer that is re-used (and so stays in the cache)?
Synthetic operands, the library's kernels as the step launches them (reducing chain; three-job weight gradient with e4m3 images).
    B=640 python tools/dbg/chunk_potential.py 640 320 128 64 32 16"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
import relationnetworks_clevr_amd as pkg
H = pkg.rn_hip
H.load()
B, n, G, L = int(os.environ.get("B", 640)), int(os.environ.get("N_OBJ", 64)), 256, 4
njp = (n + 31) // 32 * 32
chunks = [int(a) for a in sys.argv[1:]] or [B, 128, 64, 32]
ONLY = os.environ.get("ONLY", "")               # "chain" / "wgrad": that launch alone
g = torch.Generator(device="cuda").manual_seed(1)
M = B * n * njp
A8 = [(torch.rand(M, G, device="cuda", generator=g) * 2).to(torch.float8_e4m3fn) for _ in range(3)]      # H_0..2 images of the whole batch
dxg = torch.rand(B, G, device="cuda", generator=g) - 0.5
Wt = list((torch.rand(L - 1, 65536, device="cuda", generator=g) * 0.2 - 0.1).bfloat16())
dW = [torch.empty(G, G, device="cuda") for _ in range(3)]; db = [torch.empty(G, device="cuda") for _ in range(3)]


def build(C):
    """launch list for chunks of C questions"""
    Mc = C * n * njp
    masks = list(torch.randint(0, 256, (L, H.g_chain_rr_mask_bytes(Mc)), dtype=torch.uint8, device="cuda", generator=g))
    dZ = list(torch.empty(L - 2, Mc, G, dtype=torch.bfloat16, device="cuda"))           # ONE chunk-sized buffer, re-used
    tpu = H.g_chain_bwd_rr_red_tpu(Mc, n, njp)
    whole = H.g_chain_bwd_rr_red_whole(Mc, n, njp, tpu)
    rj = torch.empty(H.g_chain_bwd_rr_red_records(Mc, n, njp, tpu, whole), 32, G, device="cuda"); ri = torch.empty(Mc // 16, G, device="cuda")
    red = [None, dZ[0], dZ[1], None]

    def run():
        for c in range(B // C):
            sl = slice(c * Mc, (c + 1) * Mc)
            if ONLY != "wgrad":
                H.g_chain_bwd_rr_red(dxg[c * C:(c + 1) * C], masks, Wt, red, Mc, n, G, rj, ri, tpu, njp=njp, whole=whole)
            if ONLY != "chain":
                H.g_wgrad_blocked([(dZ[1], A8[0][sl], dW[0], db[0]), (dZ[0], A8[1][sl], dW[1], db[1]), (None, A8[2][sl], dW[2], db[2])], Mc,
                                  dxg=dxg[c * C:(c + 1) * C], rows_per_question=n * njp)
    return run


def timeit(fn, reps=7):
    for _ in range(2):
        fn()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


flops = 2.0 * B * n * n * G * G * 6
for rep in range(2):
    for C in chunks:
        if B % C:
            continue
        us = timeit(build(C))
        print(ONLY or "both", "B %d n %d, chunks of %4d questions (%4d MB of dZ per chunk): backward chain + weight gradient %8.1f us  = %6.1f us per 64 questions  (%.3f of 2.5 PF)"
              % (B, n, C, 2 * C * n * njp * G * 2 // 1000000, us, us * 64 / B, flops / (us * 1e-6) / 2.5e15), flush=True)
