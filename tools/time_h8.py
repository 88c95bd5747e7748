"""bf16 vs e4m3 copies of the stored activations at the headline shape (B=64, n=64): the two forward chains and the two
streaming weight-gradient kernels, each alone on the chip (median of 20 single launches)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import relationnetworks_clevr_amd as pkg
from bench import time_launch
H = pkg.rn_hip; H.load()
B, n, k, Q, G = 64, 64, 26, 128, 256
M = B * n * n; kt = 2 * k + Q
x = torch.randn(B, n, k, device='cuda'); q = torch.randn(B, Q, device='cuda')
Ws = [torch.randn(G, kt if l == 0 else G, device='cuda') * 0.05 for l in range(4)]
bs = [torch.randn(G, device='cuda') * 0.1 for _ in range(4)]
Wf = list(torch.empty(4, 65536, dtype=torch.bfloat16, device='cuda'))
w0T = torch.empty(kt, G, device='cuda')
H.pack_matrix_frag_many([(Ws[0], kt, 1, G, k, Wf[0], 1), (Ws[0], kt, 1, G, kt, w0T, 2)] + [(Ws[l], G, 1, G, G, Wf[l], 0) for l in range(1, 4)])
Whi = list(torch.empty(4, 65536, dtype=torch.float16, device='cuda')); Wlo = list(torch.empty(4, 65536, dtype=torch.float16, device='cuda'))
H.pack_matrix_frag_many([(Ws[l], Ws[l].shape[1], 1, G, k if l == 0 else G, Whi[l], 4 | int(l == 0)) for l in range(4)]
                        + [(Ws[l], Ws[l].shape[1], 1, G, k if l == 0 else G, Wlo[l], 8 | int(l == 0)) for l in range(4)])
Xp16 = torch.empty(B * n, 64, dtype=torch.float16, device='cuda'); Xpb = torch.empty(B * n, 64, dtype=torch.bfloat16, device='cuda')
Vc = torch.empty(B * n, G, device='cuda')
H.pair_tables(x, q, w0T, bs[0], Xp16, Vc, B, n, k, Q, G); H.pair_tables(x, q, w0T, bs[0], Xpb, Vc, B, n, k, Q, G)
masks = list(torch.empty(4, H.g_chain_rr_mask_bytes(M), dtype=torch.uint8, device='cuda'))
part = torch.empty(M // 256, G, device='cuda')
Hb = list(torch.empty(3, M, G, dtype=torch.bfloat16, device='cuda')) + [None]
H8 = list(torch.empty(3, M, G, dtype=torch.uint8, device='cuda').view(torch.float8_e4m3fn)) + [None]
dxg = torch.randn(B, G, device='cuda')
dZ = (torch.randn(M, G, device='cuda') * 1e-3).bfloat16()
dW = torch.empty(G, G, device='cuda'); db = torch.empty(G, device='cuda')
rows = [
    ("f16s chain, bf16 copies", lambda: H.g_chain_fwd_rr_f16s_alg0(Xp16, Vc, n, Whi, Wlo, bs, Hb, masks, part, M, G)),
    ("f16s chain, e4m3 copies", lambda: H.g_chain_fwd_rr_f16s_alg0(Xp16, Vc, n, Whi, Wlo, bs, H8, masks, part, M, G)),
    ("bf16 chain, bf16 copies", lambda: H.g_chain_fwd_rr_alg0(Xpb, Vc, n, Wf, bs, Hb, masks, part, M, G)),
    ("bf16 chain, e4m3 copies", lambda: H.g_chain_fwd_rr_alg0(Xpb, Vc, n, Wf, bs, H8, masks, part, M, G)),
    ("wgrad, bf16 A", lambda: H.g_linear_bwd_wgrad(dZ, G, Hb[1], G, dW, db, 0, M, G, G, G)),
    ("wgrad, e4m3 A", lambda: H.g_linear_bwd_wgrad(dZ, G, H8[1], G, dW, db, 0, M, G, G, G)),
    ("gated wgrad, bf16 A", lambda: H.g_linear_bwd_wgrad_gated(masks[3], dxg, n * n, Hb[2], G, dW, db, M, G, G)),
    ("gated wgrad, e4m3 A", lambda: H.g_linear_bwd_wgrad_gated(masks[3], dxg, n * n, H8[2], G, dW, db, M, G, G)),
]
only = sys.argv[1] if len(sys.argv) > 1 else None
for rep in range(2):
    for name, fn in rows:
        if only and only not in name:
            continue
        print("%-28s %8.1f us" % (name, 1e3 * time_launch(fn)), flush=True)
