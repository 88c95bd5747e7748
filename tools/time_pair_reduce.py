#!/usr/bin/env python3
"""Time rn_pair_reduce_bwd on the headline shape (B = 64, n = 64, G = 256, bf16): with and without Rj."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
import relationnetworks_clevr_amd as pkg
H = pkg.rn_hip
H.load()
B, n, G = 64, 64, 256
dZ = (torch.rand(B * n * n, G, device="cuda") - 0.5).bfloat16()
Rj = torch.empty(B * n, G, device="cuda"); Ri = torch.empty(B * n, G, device="cuda"); Rq = torch.empty(B, G, device="cuda")


def timeit(fn, reps=10):
    for _ in range(2):
        fn()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


print("Rj + Ri + Rq : %.1f us" % timeit(lambda: H.pair_reduce_bwd(dZ, G, Rj, Ri, Rq, 0, B, n, G)))
print("Ri + Rq only : %.1f us" % timeit(lambda: H.pair_reduce_bwd(dZ, G, None, Ri, Rq, 0, B, n, G)))
print("Rj only      : %.1f us" % timeit(lambda: H.pair_reduce_bwd(dZ, G, Rj, None, None, 0, B, n, G)))
x = torch.empty_like(dZ)
print("copy 134 MB  : %.1f us" % timeit(lambda: x.copy_(dZ)))
