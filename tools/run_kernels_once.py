"""Launch each hot-path kernel a few times at the headline shape (original-fp, B=64, n=64, M=262144, bf16) --
the target of the rocprofv3 --pmc / --kernel-trace runs whose summaries live in profiles/.  The launch set is the
training step's: tables + factored forward chain, backward chain without dZ_3, the two plain wgrads and the gated one,
the pair reduction and its tail."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import relationnetworks_clevr_amd as pkg
H = pkg.rn_hip; H.load()
which = sys.argv[1] if len(sys.argv) > 1 else "all"
mode = sys.argv[2] if len(sys.argv) > 2 else "f16s"        # forward arithmetic: "f16s" (the module default) or "bf16"
B, n, k, Q, G = 64, 64, 26, 128, 256
M = B * n * n; kt = 2 * k + Q
x = torch.randn(B, n, k, device='cuda'); q = torch.randn(B, Q, device='cuda')
Ws = [torch.randn(G, kt if l == 0 else G, device='cuda') * 0.05 for l in range(4)]
bs = [torch.randn(G, device='cuda') * 0.1 for _ in range(4)]
Wf = list(torch.empty(4, 65536, dtype=torch.bfloat16, device='cuda'))
Wtf = list(torch.empty(3, 65536, dtype=torch.bfloat16, device='cuda'))
w0T = torch.empty(kt, G, device='cuda')
H.pack_matrix_frag_many([(Ws[0], kt, 1, G, k, Wf[0], 1), (Ws[0], kt, 1, G, kt, w0T, 2)]
                        + [(Ws[l], G, 1, G, G, Wf[l], 0) for l in range(1, 4)]
                        + [(Ws[3 - s], 1, G, G, G, f, s == 0) for s, f in enumerate(Wtf)])
Xp = torch.empty(B * n, 64, dtype=torch.float16 if mode == "f16s" else torch.bfloat16, device='cuda'); Vc = torch.empty(B * n, G, device='cuda')
Whi = list(torch.empty(4, 65536, dtype=torch.float16, device='cuda')); Wlo = list(torch.empty(4, 65536, dtype=torch.float16, device='cuda'))
H.pack_matrix_frag_many([(Ws[l], Ws[l].shape[1], 1, G, k if l == 0 else G, Whi[l], 4 | int(l == 0)) for l in range(4)]
                        + [(Ws[l], Ws[l].shape[1], 1, G, k if l == 0 else G, Wlo[l], 8 | int(l == 0)) for l in range(4)])
Hs = list(torch.empty(3, M, G, dtype=torch.uint8, device='cuda').view(torch.float8_e4m3fn)) + [None]     # e4m3 copies, as the step keeps them
masks = list(torch.empty(4, H.g_chain_rr_mask_bytes(M), dtype=torch.uint8, device='cuda'))
part = torch.empty(M // 256, G, device='cuda')
dxg = torch.randn(B, G, device='cuda')
dZs = [None] + list(torch.empty(3, M, G, dtype=torch.bfloat16, device='cuda'))
dW = torch.empty(G, G, device='cuda'); dW0 = torch.empty(G, kt, device='cuda'); db = torch.empty(G, device='cuda')
Rj = torch.empty(B * n, G, device='cuda'); Ri = torch.empty(B * n, G, device='cuda'); Rq = torch.empty(B, G, device='cuda')
dx = torch.empty(B, n, k, device='cuda'); dq = torch.empty(B, Q, device='cuda')
for it in range(3):
    if which in ("all", "build"): H.pair_tables(x, q, w0T, bs[0], Xp, Vc, B, n, k, Q, G)
    if which in ("all", "chain", "chainbwd"):
        if mode == "f16s": H.g_chain_fwd_rr_f16s_alg0(Xp, Vc, n, Whi, Wlo, bs, Hs, masks, part, M, G)
        else: H.g_chain_fwd_rr_alg0(Xp, Vc, n, Wf, bs, Hs, masks, part, M, G)
    if which in ("all", "bwd", "chainbwd"): H.g_chain_bwd_rr(dxg, masks, Wtf, dZs, M, n * n, G)
    if which in ("all", "wgrad"):
        H.g_linear_bwd_wgrad(dZs[1], G, Hs[1], G, dW, db, 0, M, G, G, G)
        H.g_linear_bwd_wgrad_gated(masks[3], dxg, n * n, Hs[2], G, dW, db, M, G, G)
    if which in ("all", "reduce"):
        H.pair_reduce_bwd(dZs[3], G, Rj, Ri, Rq, 0, B, n, G)
        H.pair_dx_dq(Rj, Ri, Rq, Ws[0], dx, dq, B, n, k, Q, G)
        H.wgrad0_from_reductions(Rj, Ri, Rq, x, q, dW0, db)
torch.cuda.synchronize()
