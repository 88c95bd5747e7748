#!/usr/bin/env python3
"""Time the reducing backward chain alone (B= / N_OBJ= in the environment; default the headline shape) and, with RN_DIAG=1 (a
diagnostics build: RN_LIB=<variant .so> or the in-tree library built with RN_DIAG=1), its timing ablations -- results are WRONG
in those, only the durations mean something.  Round-robin with a low-power spin in front of every launch (tools/time_fwd_f16s.py)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
import relationnetworks_clevr_amd as pkg
H = pkg.rn_hip
if os.environ.get("RN_LIB"):
    H.LIB_PATH = os.path.abspath(os.environ["RN_LIB"])
lib = H.load()
from tools.time_fwd_f16s_lib import time_variants
B, n, G, L = int(os.environ.get("B", 64)), int(os.environ.get("N_OBJ", 64)), 256, 4
njp = (n + 31) // 32 * 32
M = B * n * njp
g = torch.Generator(device="cuda").manual_seed(1)
masks = list(torch.randint(0, 256, (L, H.g_chain_rr_mask_bytes(M)), dtype=torch.uint8, device="cuda", generator=g))
dxg = torch.rand(B, G, device="cuda", generator=g) - 0.5
Wt = list((torch.rand(L - 1, 65536, device="cuda", generator=g) * 0.2 - 0.1).bfloat16())
dZ = list(torch.empty(L - 2, M, G, dtype=torch.bfloat16, device="cuda"))
tpu = H.g_chain_bwd_rr_red_tpu(M, n, njp)
whole = H.g_chain_bwd_rr_red_whole(M, n, njp, tpu)
rj = torch.empty(H.g_chain_bwd_rr_red_records(M, n, njp, tpu, whole), 32, G, device="cuda"); ri = torch.empty(M // 16, G, device="cuda")
red = [None, dZ[0], dZ[1], None]
run = lambda: H.g_chain_bwd_rr_red(dxg, masks, Wt, red, M, n, G, rj, ri, tpu, njp=njp, whole=whole)
flops = 2.0 * B * n * n * G * 3 * G
variants = {"baseline": (lambda: None, run)}
if os.environ.get("RN_DIAG", "0") == "1":
    names = {128: "tile prologue once per workgroup only", 142: "no prologue, masks, stores, waits, barriers", 2: "no mask loads", 4: "no dZ stores", 6: "no mask loads, no dZ stores", 8: "no counted waits / barriers", 14: "no masks, stores, waits, barriers",
             32: "dZ stores to L2-resident addresses", 64: "plain instead of non-temporal dZ stores"}
    for abl, what in names.items():
        variants["ABL %3d (%s)" % (abl, what)] = ((lambda a=abl: lib.rn_diag_set_abl(a)), run)
    variants["baseline again"] = (lambda: lib.rn_diag_set_abl(0), run)
res = time_variants(variants)
for k_, us in res.items():
    print("%-52s %7.1f us  (%.3f of 2.5 PF algorithmic)" % (k_, us, flops / (us * 1e-6) / 2.5e15))
