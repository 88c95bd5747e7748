"""Backward chain with / without the in-chain pair reduction at the headline shape, each alone on the chip, + the finishing kernels."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import relationnetworks_clevr_amd as pkg
from bench import time_launch
H = pkg.rn_hip; H.load()
B, n, G, L = 64, 64, 256, 4
M = B * n * n
g = torch.Generator(device="cuda").manual_seed(1)
masks = list(torch.randint(0, 256, (L, H.g_chain_rr_mask_bytes(M)), dtype=torch.uint8, device="cuda", generator=g))
dxg = torch.rand(B, G, device="cuda", generator=g) - 0.5
Wt = list(torch.empty(L - 1, 65536, dtype=torch.bfloat16, device="cuda"))
for st in range(L - 1):
    H.pack_matrix_frag((torch.rand(G, G, device="cuda", generator=g) - 0.5) * 0.3, 1, G, G, G, Wt[st], st == 0)
full = [None] + list(torch.zeros(L - 1, M, G, dtype=torch.bfloat16, device="cuda"))
part = [None] + list(torch.zeros(L - 2, M, G, dtype=torch.bfloat16, device="cuda")) + [None]
rj = torch.empty(H.chain_reduce_part_bytes(M, 0) // 4, device="cuda"); ri = torch.empty(H.chain_reduce_part_bytes(M, 1) // 4, device="cuda")
Rj = torch.empty(B * n, G, device="cuda"); Ri = torch.empty(B * n, G, device="cuda"); Rq = torch.empty(B, G, device="cuda")
rows = [("bwd chain, dZ_0 stored", lambda: H.g_chain_bwd_rr(dxg, masks, Wt, full, M, n * n, G)),
        ("bwd chain, reduced on chip", lambda: H.g_chain_bwd_rr_reduce(dxg, masks, Wt, part, rj, ri, n, M, G)),
        ("pair_reduce_bwd (dZ_0 pass + finish)", lambda: H.pair_reduce_bwd(full[3], G, Rj, Ri, Rq, H.RN_BF16, B, n, G)),
        ("pair_reduce_from_chain", lambda: H.pair_reduce_from_chain(rj, ri, Rj, Ri, Rq, B, n, G))]
for rep in range(2):
    for name, fn in rows:
        print("%-40s %8.1f us" % (name, 1e3 * time_launch(fn)), flush=True)
