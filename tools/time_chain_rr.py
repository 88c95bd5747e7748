#!/usr/bin/env python3
"""Time the forward chains on the headline shape (M = 64 * 4096 pair rows): old LDS-resident kernel vs the
register-resident one, with and without activation stores."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
import relationnetworks_clevr_amd as pkg
H = pkg.rn_hip
H.load()
M, G, K0 = 64 * 4096, 256, 192
torch.manual_seed(0)
P = (torch.rand(M, K0, device="cuda") * 2 - 1).bfloat16()
Ws = [(torch.rand(G, K0 if l == 0 else G, device="cuda") - 0.5) * 0.3 for l in range(4)]
bs = [(torch.rand(G, device="cuda") - 0.5) * 0.6 for _ in range(4)]
Wp = [w.bfloat16().contiguous() for w in Ws]
Wf = []
for l, w in enumerate(Ws):
    f = torch.empty(65536, dtype=torch.bfloat16, device="cuda")
    H.pack_matrix_frag(w, w.shape[1], 1, G, w.shape[1], f, l == 0)
    Wf.append(f)
Hs = [torch.empty(M, G, dtype=torch.bfloat16, device="cuda") for _ in range(4)]
masks = list(torch.zeros(4, H.g_chain_rr_mask_bytes(M), dtype=torch.uint8, device="cuda"))
dxg = torch.rand(64, G, device="cuda") - 0.5
dZs = list(torch.empty(4, M, G, dtype=torch.bfloat16, device="cuda"))
Wts = [Ws[3 - s].t().contiguous().bfloat16() for s in range(3)]
Wtf = list(torch.empty(3, 65536, dtype=torch.bfloat16, device="cuda"))
for s, f in enumerate(Wtf):
    H.pack_matrix_frag(Ws[3 - s], 1, G, G, G, f, s == 0)
part_old = torch.empty(M // H.g_chain_tile(), G, dtype=torch.float32, device="cuda")
part_rr = torch.empty(M // 32, G, dtype=torch.float32, device="cuda")


def timeit(fn, n=20):
    """median single-launch duration (event bracket per launch: host launch cost does not count)"""
    for _ in range(3):
        fn()
    ts = []
    for _ in range(n):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


flops = 2.0 * M * G * (K0 + 3 * G)
for name, fn in [
    ("old  store", lambda: H.g_chain_fwd(P, K0, Wp, bs, Hs, [K0, G, G, G], part_old, 0, M, G)),
    ("old  nostore", lambda: H.g_chain_fwd(P, K0, Wp, bs, [None] * 4, [K0, G, G, G], part_old, 0, M, G)),
    ("rr   all H", lambda: H.g_chain_fwd_rr(P, K0, Wf, bs, Hs, None, K0, part_rr, M, G)),
    ("rr   train (H0-2 + masks)", lambda: H.g_chain_fwd_rr(P, K0, Wf, bs, Hs[:3] + [None], masks, K0, part_rr, M, G)),
    ("rr   infer", lambda: H.g_chain_fwd_rr(P, K0, Wf, bs, None, None, K0, part_rr, M, G)),
    ("old  bwd", lambda: H.g_chain_bwd(Hs[3], dxg, Wts, [Hs[2], Hs[1], Hs[0]], dZs, 0, M, 4096, G)),
    ("rr   bwd", lambda: H.g_chain_bwd_rr(dxg, masks, Wtf, dZs, M, 4096, G)),
]:
    us = timeit(fn)
    print("%-28s %8.1f us   %7.1f TFLOP/s" % (name, us, (flops * (0.79 if "bwd" in name else 1.0)) / us * 1e-6))
for abl in ():
    os.environ["RN_RR_ABL"] = str(abl)
    us = timeit(lambda: H.g_chain_bwd_rr(dxg, masks, Wtf, dZs, M, 4096, G))
    print("rr bwd ablation %2d (1=no prologue 2=no gate loads 4=no stores 8=no sync): %8.1f us" % (abl, us))
