#!/usr/bin/env python3
"""Time rn_g_linear_bwd_wgrad on the headline shape (M = 64 * 4096): general kernel (RN_WGRAD_V1=1) vs streaming kernel."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
os.environ["RN_WGRAD_STREAM_192"] = "1"
import relationnetworks_clevr_amd as pkg
H = pkg.rn_hip
H.load()
M, N = 64 * 4096, 256


def timeit(fn, n=10):
    for _ in range(2):
        fn()
    ts = []
    for _ in range(n):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


for K, Kt in ((256, 256), (192, 180)):
    dZ = (torch.rand(M, N, device="cuda") - 0.5).bfloat16()
    A = (torch.rand(M, K, device="cuda") - 0.5).bfloat16()
    dW = torch.empty(N, Kt, device="cuda"); db = torch.empty(N, device="cuda")
    for v1 in ("1", "0"):
        os.environ["RN_WGRAD_V1"] = v1
        us = timeit(lambda: H.g_linear_bwd_wgrad(dZ, N, A, K, dW, db, 0, M, N, K, Kt))
        gb = (M * N + M * K) * 2 / 1e9
        print("K=%d %s: %7.1f us  (operands %.0f MB -> %.2f TB/s)" % (K, "general  " if v1 == "1" else "streaming", us, gb * 1e3, gb / us * 1e3))
    os.environ["RN_WGRAD_ABL"] = "1"
    us = timeit(lambda: H.g_linear_bwd_wgrad(dZ, N, A, K, dW, db, 0, M, N, K, Kt))
    print("K=%d streaming, stream only: %7.1f us" % (K, us))
    os.environ["RN_WGRAD_ABL"] = "2"
    us = timeit(lambda: H.g_linear_bwd_wgrad(dZ, N, A, K, dW, db, 0, M, N, K, Kt))
    print("K=%d streaming, compute only: %7.1f us" % (K, us))
    os.environ.pop("RN_WGRAD_ABL")
