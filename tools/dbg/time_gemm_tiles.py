#!/usr/bin/env python3
"""rn_g_linear_fwd (fp32 / bf16 / bf16x3) with the 64 x 64 and the 128 x 256 workgroup tiles over M: where the switch belongs
(rn_debug_gemm_small_below moves it; profiles/r05_ablations/gemm_tiles.txt)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch, relationnetworks_clevr_amd as pkg
H = pkg.rn_hip
lib = H.load()
for nk, code in ((512, H.RN_F32), (512, H.RN_F32X3), (256, H.RN_F32), (256, H.RN_BF16)):
    for tag, v in (("small", 1 << 30), ("big", 0)):
        lib.rn_debug_gemm_small_below(v)
        N = K = nk
        dt = torch.bfloat16 if code == H.RN_BF16 else torch.float32
        for M in (576, 2304, 9216, 18432, 36864, 73728, 147456):
            A = torch.randn(M, K, device="cuda").to(dt); W = torch.randn(N, K, device="cuda").to(dt); b = torch.randn(N, device="cuda"); out = torch.empty(M, N, device="cuda", dtype=dt)
            for _ in range(5): H.g_linear_fwd(A, K, W, K, b, out, N, code, M, N, K)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50): H.g_linear_fwd(A, K, W, K, b, out, N, code, M, N, K)
            e1.record(); torch.cuda.synchronize()
            print("%-5s N=K=%d code=%d M=%6d  %7.1f us" % (tag, N, code, M, e0.elapsed_time(e1) * 20))
lib.rn_debug_gemm_small_below(-1)
