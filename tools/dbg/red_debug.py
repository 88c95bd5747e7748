"""Debug: error pattern of the reducing backward chain (rows / features)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import numpy as np, torch
import relationnetworks_clevr_amd as pkg
from oracle import formula
import test_gpu_kernels as T
H = pkg.rn_hip; H.load()
B, n, tpu = 2, 64, int(sys.argv[1]) if len(sys.argv) > 1 else 1
L, G = 4, 256
M = B * n * n
g = torch.Generator(device="cuda").manual_seed(7)
masks = list(torch.randint(0, 256, (L, H.g_chain_rr_mask_bytes(M)), dtype=torch.uint8, device="cuda", generator=g))
if len(sys.argv) > 2 and sys.argv[2] == "ones":
    masks[0].fill_(255)
dxg = (torch.rand(B, G, device="cuda", generator=g) - 0.5)
Wt = list(torch.empty(L - 1, 65536, dtype=torch.bfloat16, device="cuda"))
Ws = []
for st in range(L - 1):
    Ws.append(T.bf16_round(formula.hash_uniform((G, G), 340 + st, -0.15, 0.15)))
    H.pack_matrix_frag(T.dev(Ws[-1]), 1, G, G, G, Wt[st], st == 0)
new = [None] + list(torch.zeros(L - 2, M, G, dtype=torch.bfloat16, device="cuda")) + [None]
nu = (n // 8) // tpu
rj_part = torch.full((M // 256 // tpu, 32, G), float("nan"), device="cuda")
ri_part = torch.full((M // 16, G), float("nan"), device="cuda")
H.g_chain_bwd_rr_red(dxg, masks, Wt, new, M, n, G, rj_part, ri_part, tpu)
Rj = torch.empty(B * n, G, device="cuda"); Ri = torch.empty(B * n, G, device="cuda"); Rq = torch.empty(B, G, device="cuda")
H.pair_reduce_parts(rj_part, ri_part, Rj, Ri, Rq, B, n, G, nu)
torch.cuda.synchronize()
gate0 = torch.from_numpy(T.rr_mask_decode(masks[0], M, 0)).cuda()
dz2 = T.unblock(new[2]).double()
dz0 = ((dz2 @ T.dev(Ws[2]).double()) * gate0).view(B, n, n, G)
ej = (Rj.double().view(B, n, G) - dz0.sum(1)).abs()
ei = (Ri.double().view(B, n, G) - dz0.sum(2)).abs()
print("scale", dz0.abs().max().item(), "nan parts", torch.isnan(rj_part).sum().item(), torch.isnan(ri_part).sum().item())
print("Rj err by j (b=0):", np.array2string(ej[0].max(1).values.cpu().numpy(), precision=1, max_line_width=200))
print("Rj err by feature%32 (b=0):", np.array2string(ej[0].view(n, 8, 32).amax((0, 1)).cpu().numpy(), precision=1, max_line_width=200))
print("Rj err by block (b=0):", np.array2string(ej[0].view(n, 8, 32).amax((0, 2)).cpu().numpy(), precision=1, max_line_width=200))
print("Ri err by i (b=0):", np.array2string(ei[0].max(1).values.cpu().numpy(), precision=1, max_line_width=200))
print("Ri err by block (b=0):", np.array2string(ei[0].view(n, 8, 32).amax((0, 2)).cpu().numpy(), precision=1, max_line_width=200))
print("Rq err", (Rq.double() - dz0.sum((1, 2))).abs().max().item())
# which actual row of unit 0 matches which expected row?
exp = dz0[0, 0:8 * tpu, 0:32, :].sum(0)            # (32, G)
act = rj_part[0].double()
d = (act[:, None, :] - exp[None, :, :]).abs().amax(2)       # (actual row, expected row)
print("unit 0: actual row -> best expected row (err):", [(int(d[r].argmin()), float("%.1e" % d[r].min())) for r in range(32)])
# wave-level Ri partial of wave-tile 0: expected sum over j<32 of dz0[0,0,j,:]
e_ri = dz0[0, 0, 0:32, :].sum(0)
print("ri_part[0] err by block:", (ri_part[0].double() - e_ri).abs().view(8, 32).amax(1).cpu().numpy())
# per-feature-in-block pattern of unit 0 row 1
print("row1 err by feature in block0:", np.array2string((act[1] - exp[1]).abs()[:32].cpu().numpy(), precision=1, max_line_width=250))
rows = torch.arange(32)
h0 = ((rows >> 2) & 1) == 0
A = dz0[0, 0, 0:32][h0.cuda()].sum(0); Bs = dz0[0, 0, 0:32][~h0.cuda()].sum(0)
got = ri_part[0].double()
for name, cand in (("A+B", A + Bs), ("2A", 2 * A), ("2B", 2 * Bs), ("A", A), ("B", Bs)):
    print("ri_part[0] vs", name, (got - cand).abs().max().item())
# group-wise: which groups j contribute?
for jj in range(4):
    sel = ((rows >> 3) == jj).cuda()
    print("group", jj, "only:", (got - dz0[0, 0, 0:32][sel].sum(0)).abs().max().item())
