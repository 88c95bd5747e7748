"""Launch each hot-path kernel a few times at the headline shape (original-fp, B=64, n=64, M=262144, bf16) --
the target of the rocprofv3 --pmc / --kernel-trace runs whose summaries live in profiles/."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import relationnetworks_clevr_amd as pkg
H = pkg.rn_hip; H.load()
which = sys.argv[1] if len(sys.argv) > 1 else "all"
B, n, k, Q, G = 64, 64, 26, 128, 256
M = B * n * n; K0 = 192
x = torch.randn(B, n, k, device='cuda'); q = torch.randn(B, Q, device='cuda')
P = torch.empty(M, K0, dtype=torch.bfloat16, device='cuda')
Ws = [torch.randn(G, 180 if l == 0 else G, device='cuda') * 0.05 for l in range(4)]
bs = [torch.randn(G, device='cuda') * 0.1 for _ in range(4)]
Wf = list(torch.empty(4, 65536, dtype=torch.bfloat16, device='cuda'))
Wtf = list(torch.empty(3, 65536, dtype=torch.bfloat16, device='cuda'))
H.pack_matrix_frag_many([(w, w.shape[1], 1, G, w.shape[1], f, l == 0) for l, (w, f) in enumerate(zip(Ws, Wf))]
                        + [(Ws[3 - s], 1, G, G, G, f, s == 0) for s, f in enumerate(Wtf)])
Hs = list(torch.empty(3, M, G, dtype=torch.bfloat16, device='cuda')) + [None]
masks = list(torch.empty(4, H.g_chain_rr_mask_bytes(M), dtype=torch.uint8, device='cuda'))
part = torch.empty(M // 32, G, device='cuda')
dxg = torch.randn(B, G, device='cuda')
dZs = list(torch.empty(4, M, G, dtype=torch.bfloat16, device='cuda'))
dW = torch.empty(G, G, device='cuda'); dW0 = torch.empty(G, 180, device='cuda'); db = torch.empty(G, device='cuda')
Rj = torch.empty(B * n, G, device='cuda'); Ri = torch.empty(B * n, G, device='cuda'); Rq = torch.empty(B, G, device='cuda')
for it in range(3):
    if which in ("all", "build"): H.pair_build_fwd(x, q, P, 0, B, n, k, Q, K0)
    if which in ("all", "chain"): H.g_chain_fwd_rr(P, K0, Wf, bs, Hs, masks, K0, part, M, G)
    if which in ("all", "bwd"): H.g_chain_bwd_rr(dxg, masks, Wtf, dZs, M, n * n, G)
    if which in ("all", "wgrad"):
        H.g_linear_bwd_wgrad(dZs[0], G, Hs[2], G, dW, db, 0, M, G, G, G)
        H.g_linear_bwd_wgrad(dZs[3], G, P, K0, dW0, db, 0, M, G, K0, 180)
    if which in ("all", "reduce"): H.pair_reduce_bwd(dZs[3], G, Rj, Ri, Rq, 0, B, n, G)
torch.cuda.synchronize()
