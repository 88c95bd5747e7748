#!/usr/bin/env python3
"""Isolated durations (median of 20 single launches, HIP events) of the small kernels around the chains at the headline shape."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import relationnetworks_clevr_amd as pkg
from bench import time_launch
H = pkg.rn_hip; H.load()
B, n, k, Q, G, A = 64, 64, 26, 128, 256, 28
M = B * n * n; kt = 2 * k + Q
dev = "cuda"
x = torch.randn(B, n, k, device=dev); q = torch.randn(B, Q, device=dev)
W0 = torch.randn(G, kt, device=dev) * 0.05; b0 = torch.randn(G, device=dev) * 0.1
w0T = W0.t().contiguous()
Xp = torch.empty(B * n, 64, dtype=torch.float16, device=dev); Vc = torch.empty(B * n, G, device=dev)
part = torch.randn(M // 256, G, device=dev); xg = torch.empty(B, G, device=dev)
fw = [torch.randn(256, 256, device=dev) * 0.05, torch.randn(256, 256, device=dev) * 0.05, torch.randn(A, 256, device=dev) * 0.05]
fT = [w.t().contiguous() for w in fw]; fb = [torch.randn(256, device=dev) * 0.1, torch.randn(256, device=dev) * 0.1, torch.randn(A, device=dev) * 0.1]
f1 = torch.empty(B, 256, device=dev); f2 = torch.empty(B, 256, device=dev); out = torch.empty(B, A, device=dev); loss = torch.empty((), device=dev)
label = torch.randint(0, A, (B,), device=dev); gloss = torch.ones((), device=dev)
dW = (torch.empty(256, 256, device=dev), torch.empty(256, 256, device=dev), torch.empty(A, 256, device=dev))
db = (torch.empty(256, device=dev), torch.empty(256, device=dev), torch.empty(A, device=dev)); dxg = torch.empty(B, G, device=dev)
dZ = torch.randn(M, G, device=dev).bfloat16(); Hh = torch.randn(M, G, device=dev).abs().to(torch.float8_e4m3fn)     # e4m3 copies, as the step keeps them
Rj = torch.empty(B * n, G, device=dev); Ri = torch.empty(B * n, G, device=dev); Rq = torch.empty(B, G, device=dev)
dx = torch.empty(B, n, k, device=dev); dq = torch.empty(B, Q, device=dev); dW0 = torch.empty(G, kt, device=dev); db0 = torch.empty(G, device=dev)
gW = torch.empty(G, G, device=dev); gB = torch.empty(G, device=dev)
masks = torch.randint(0, 255, (H.g_chain_rr_mask_bytes(M),), dtype=torch.uint8, device=dev)
dZb = [H.rows_to_blocked(dZ) for _ in range(2)]; H8 = [H.rows_to_blocked(Hh) for _ in range(3)]; H.relu_gate_image(masks, H8[2], M)
gWs = [torch.empty(G, G, device=dev) for _ in range(3)]; gBs = [torch.empty(G, device=dev) for _ in range(3)]
jobs = [(dZb[0], H8[0], gWs[0], gBs[0]), (dZb[1], H8[1], gWs[1], gBs[1]), (None, H8[2], gWs[2], gBs[2])]
mask_fp = (torch.rand(B, 256, device=dev) > 0.5).float() * 2; sync_fp = H.f_phi_split_sync_ws(dev)
ws_fp = H.f_phi_split(part[:B * 16], 16, xg, fw, fb, fT, mask_fp, label, f1, f2, out, loss, sync_fp, dxg=dxg)
part32 = torch.randn(M // 32 * (1 if n % 32 == 0 else 2), G, device=dev)
rows = [
    ("pair_tables", lambda: H.pair_tables(x, q, w0T, b0, Xp, Vc, B, n, k, Q, G)),
    ("pair_sum_fwd (segsum of the chain partials)", lambda: H.pair_sum_fwd(part, G, xg, H.RN_F32, B, (n * n) // 256, G)),
    ("f_phi_fwd_nll", lambda: H.f_phi_fwd_nll(xg, fT, fb, None, label, f1, f2, out, loss, transposed=True)),
    ("f_phi_fwd_bwd_from_partials (row-split, fwd + loss + dz chain)", lambda: H.f_phi_fwd_bwd_from_partials(part[:B * 16], 16, xg, fT, fb, fw, mask_fp, label, f1, f2, out, loss, dxg)),
    ("f_phi_split (feature-split MFMA, fwd + loss + dz chain)", lambda: H.f_phi_split(part[:B * 16], 16, xg, fw, fb, fT, mask_fp, label, f1, f2, out, loss, sync_fp, dxg=dxg)),
    ("f_phi_split, forward + loss only", lambda: H.f_phi_split(part[:B * 16], 16, xg, fw, fb, fT, mask_fp, label, f1, f2, out, loss, sync_fp)),
    ("f_phi_bwd_grads (six parameter gradients)", lambda: H.f_phi_bwd_grads(ws_fp, xg, f1, f2, dW, db)),
    ("f_phi_bwd_nll (dz + grads)", lambda: H.f_phi_bwd_nll(gloss, label, out, f2, f1, xg, fw, None, dW, db, dxg)),
    ("pair_reduce_bwd (+finish)", lambda: H.pair_reduce_bwd(dZ, G, Rj, Ri, Rq, H.RN_BF16, B, n, G)),
    ("pair_dx_dq", lambda: H.pair_dx_dq(Rj, Ri, Rq, W0, dx, dq, B, n, k, Q, G)),
    ("wgrad0_from_reductions (part + finish)", lambda: H.wgrad0_from_reductions(Rj, Ri, Rq, x, q, dW0, db0)),
    ("g_wgrad_blocked, step's three jobs (2 stored + gate)", lambda: H.g_wgrad_blocked(jobs, M, dxg=dxg, rows_per_question=n * n)),
    ("pair_sum_tiles", lambda: H.pair_sum_tiles(part32, xg, M, n * n, G)),
]
only = sys.argv[1] if len(sys.argv) > 1 else None
for name, fn in rows:
    if only and only not in name:
        continue
    print("%-48s %8.1f us" % (name, 1e3 * time_launch(fn)))
