#!/usr/bin/env python3
"""Time the backward chain on the headline shape (M = 64 * 4096; B= / N_OBJ= in the environment for others, e.g. B=32 N_OBJ=196:
random masks, so on a padded j axis only the TIMES mean something), alone on the chip: the reducing variant (rn_g_chain_bwd_rr_red
+ rn_pair_reduce_parts), with and without the balanced tail, against the stored-dZ_0 one (rn_g_chain_bwd_rr + rn_pair_reduce_bwd).
Optional argv[1]: another library."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
import relationnetworks_clevr_amd as pkg
H = pkg.rn_hip
if len(sys.argv) > 1:
    H.LIB_PATH = os.path.abspath(sys.argv[1])
H.load()
from bench import time_launch
B, n, G, L = int(os.environ.get("B", 64)), int(os.environ.get("N_OBJ", 64)), 256, 4
njp = (n + 31) // 32 * 32
M = B * n * njp
g = torch.Generator(device="cuda").manual_seed(1)
masks = list(torch.randint(0, 256, (L, H.g_chain_rr_mask_bytes(M)), dtype=torch.uint8, device="cuda", generator=g))
dxg = torch.rand(B, G, device="cuda", generator=g) - 0.5
Wt = list((torch.rand(L - 1, 65536, device="cuda", generator=g) * 0.2 - 0.1).bfloat16())
dZ = list(torch.empty(L - 1, M, G, dtype=torch.bfloat16, device="cuda"))
tpu = H.g_chain_bwd_rr_red_tpu(M, n, njp)
units, whole = H.g_chain_bwd_rr_red_units(M, n, njp, tpu), H.g_chain_bwd_rr_red_whole(M, n, njp, tpu)
rj = torch.empty(H.g_chain_bwd_rr_red_records(M, n, njp, tpu, whole), 32, G, device="cuda"); ri = torch.empty(M // 16, G, device="cuda")
Rj = torch.empty(B * n, G, device="cuda"); Ri = torch.empty(B * n, G, device="cuda"); Rq = torch.empty(B, G, device="cuda")
red = [None, dZ[0], dZ[1], None]
old = [None, dZ[0], dZ[1], dZ[2]]
nu = ((n + 7) // 8) // tpu
print("B=%d n=%d njp=%d: %d units of %d tiles, %d whole + %d single tiles" % (B, n, njp, units, tpu, whole, (units - whole) * tpu))
for wh, name in ((whole, "balanced tail"), (units, "whole units  ")):
    print("backward chain, reducing, %s  : %7.1f us" % (name, 1e3 * time_launch(lambda: H.g_chain_bwd_rr_red(dxg, masks, Wt, red, M, n, G, rj, ri, tpu, njp=njp, whole=wh))))
    print("  + partial sums -> Rj, Ri, Rq          : %7.1f us" % (1e3 * time_launch(lambda: H.pair_reduce_parts(rj, ri, Rj, Ri, Rq, B, n, G, nu, njp=njp, tpu=tpu, whole=wh))))
print("backward chain, dZ_0 stored              : %7.1f us" % (1e3 * time_launch(lambda: H.g_chain_bwd_rr(dxg, masks, Wt, old, M, n * njp, G))))
print("  + rn_pair_reduce_bwd                   : %7.1f us" % (1e3 * time_launch(lambda: H.pair_reduce_bwd(dZ[2], G, Rj, Ri, Rq, 0, B, n, G, njp=njp))))
