"""Decode what rn_f_phi_split's first layer computes: identity weights, xg[r][k] = 1000 r + k (relu keeps them)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
import relationnetworks_clevr_amd as pkg
H = pkg.rn_hip; H.load()
B, G, A = 64, 256, 28
dev = "cuda"
xg = (torch.arange(B, device=dev)[:, None] * 1000.0 + torch.arange(G, device=dev)[None, :]).float()
I = torch.eye(G, device=dev)
fw = [I.clone(), I.clone(), torch.zeros(A, G, device=dev)]
fb = [torch.zeros(G, device=dev), torch.zeros(G, device=dev), torch.zeros(A, device=dev)]
f1 = torch.full((B, G), -1.0, device=dev); f2 = torch.full((B, G), -1.0, device=dev); out = torch.empty(B, A, device=dev)
sync = H.f_phi_split_sync_ws(dev)
H.f_phi_split(None, 0, xg, fw, fb, None, None, None, f1, f2, out, None, sync)
torch.cuda.synchronize()
print("status", H.f_phi_split_status())
print("f1 == xg:", bool(torch.equal(f1, xg)), " f2 == xg:", bool(torch.equal(f2, xg)))
bad = (f1 != xg).nonzero()
print("mismatches", bad.shape[0])
for r, c in bad[:12].tolist():
    v = f1[r, c].item()
    print("f1[%d][%d] = %.1f  (that is xg[%d][%d])" % (r, c, v, int(v) // 1000, int(v) % 1000))
