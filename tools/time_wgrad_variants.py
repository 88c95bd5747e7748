"""Stand-alone timing of the streaming wgrad variants at the headline shape (plain / gated)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import relationnetworks_clevr_amd as pkg
H = pkg.rn_hip; H.load()
B, n, G = 64, 64, 256
M = B * n * n
g = torch.Generator(device="cuda").manual_seed(5)
mask = torch.randint(0, 256, (H.g_chain_rr_mask_bytes(M),), dtype=torch.uint8, device="cuda", generator=g)
dxg = torch.rand(B, G, device="cuda") - 0.5
A = (torch.rand(M, G, device="cuda") - 0.5).bfloat16()
dZ = (torch.rand(M, G, device="cuda") - 0.5).bfloat16()
def tm(f, nrep=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(nrep): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / nrep * 1e3
res = {}
for name, env in (("plain", "0"),):
    os.environ["RN_WGRAD_PIPE"] = env
    dW = torch.empty(G, G, device="cuda"); db = torch.empty(G, device="cuda")
    print("%-10s %.1f us" % (name, tm(lambda: H.g_linear_bwd_wgrad(dZ, G, A, G, dW, db, H.RN_BF16, M, G, G, G))))
    res[name] = (dW.clone(), db.clone())
dW = torch.empty(G, G, device="cuda"); db = torch.empty(G, device="cuda")
print("%-10s %.1f us" % ("gated", tm(lambda: H.g_linear_bwd_wgrad_gated(mask, dxg, n * n, A, G, dW, db, M, G, G))))
