#!/usr/bin/env python3
"""Launch the headline forward chain and the reducing backward chain N times each (argv[1], default 100) -- a target for rocprofv3
(PC sampling, counters).  Random weights / masks: only the kernels' behaviour in time means something."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
import relationnetworks_clevr_amd as pkg
H = pkg.rn_hip
H.load()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
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
dxg = torch.rand(B, G, device="cuda") - 0.5
Wt = list((torch.rand(L - 1, 65536, device="cuda") * 0.2 - 0.1).bfloat16())
dZ = list(torch.empty(L - 2, M, G, dtype=torch.bfloat16, device="cuda"))
tpu = H.g_chain_bwd_rr_red_tpu(M, n, n)
whole = H.g_chain_bwd_rr_red_whole(M, n, n, tpu)
rj = torch.empty(H.g_chain_bwd_rr_red_records(M, n, n, tpu, whole), 32, G, device="cuda"); ri = torch.empty(M // 16, G, device="cuda")
which = os.environ.get("WHICH", "fb")
for _ in range(N):
    if "f" in which:
        H.g_chain_fwd_rr_f16s_alg0(Xp, Vc, n, hi, lo, bs, Hs, masks, part, M, G)
    if "b" in which:
        H.g_chain_bwd_rr_red(dxg, masks, Wt, [None, dZ[0], dZ[1], None], M, n, G, rj, ri, tpu, whole=whole)
torch.cuda.synchronize()
print("done", N)
