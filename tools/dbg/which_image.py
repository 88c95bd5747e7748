import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
import relationnetworks_clevr_amd as pkg
H = pkg.rn_hip; H.load()
B, n, L, G, k, Q = 2, 64, 4, 256, 26, 128
M, kt = B * n * n, 180
torch.manual_seed(0)
x = torch.rand(B, n, k, device="cuda"); q = torch.rand(B, Q, device="cuda")
Ws = [torch.rand(G, kt if l == 0 else G, device="cuda") * 0.05 for l in range(L)]
bs = [torch.zeros(G, device="cuda") for _ in range(L)]
V = 4
hi = [torch.empty(65536, dtype=torch.float16, device="cuda")] + [torch.empty(V, 65536, dtype=torch.float16, device="cuda") for _ in range(1, L)]
lo = torch.empty(65536, dtype=torch.float16, device="cuda"); w0T = torch.empty(kt, G, device="cuda")
jobs = [(Ws[0], kt, 1, G, k, hi[0], 5), (Ws[0], kt, 1, G, k, lo, 9), (Ws[0], kt, 1, G, kt, w0T, 2)] + [(Ws[l], G, 1, G, G, hi[l], 4 | (V << 8)) for l in range(1, L)]
H.pack_matrix_frag_many(jobs)
# make layer-3 image d = (d + 1) x image 0: the pair-sum partial of tile t then scales with (t % 4 + 1)
for d in range(V):
    hi[3][d] = (hi[3][0].float() * (d + 1)).half() if d else hi[3][0]
for l in (1, 2):
    for d in range(1, V):
        hi[l][d] = hi[l][0]
Xp = torch.empty(B * n, 64, dtype=torch.float16, device="cuda"); Vc = torch.empty(B * n, G, device="cuda")
H.pair_tables(x, q, w0T, bs[0], Xp, Vc, B, n, k, Q, G)
part = torch.empty(M // 256, G, device="cuda")
H.g_chain_fwd_rr_f16s_alg0(Xp, Vc, n, hi, lo, bs, None, None, part, M, G)
torch.cuda.synchronize()
s = part.sum(1)
print((s / s[0]).tolist())
