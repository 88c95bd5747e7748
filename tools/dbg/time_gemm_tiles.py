#!/usr/bin/env python3
"""rn_g_linear_fwd (fp32 / bf16) with the 64 x 64 and the 128 x 256 workgroup tiles over M: where the switch belongs
(RN_GEMM_SMALL_BELOW is read once per process: one child process per setting)."""
import os, subprocess, sys
if len(sys.argv) > 1:
    ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
    import torch, relationnetworks_clevr_amd as pkg
    H = pkg.rn_hip
    N = K = int(sys.argv[2]); code = int(sys.argv[3])
    dt = torch.bfloat16 if code == 0 else torch.float32
    for M in (576, 2304, 9216, 18432, 36864, 73728):
        A = torch.randn(M, K, device="cuda").to(dt); W = torch.randn(N, K, device="cuda").to(dt); b = torch.randn(N, device="cuda"); out = torch.empty(M, N, device="cuda", dtype=dt)
        for _ in range(5): H.g_linear_fwd(A, K, W, K, b, out, N, code, M, N, K)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): H.g_linear_fwd(A, K, W, K, b, out, N, code, M, N, K)
        e1.record(); torch.cuda.synchronize()
        print("%s N=K=%d code=%d M=%6d  %7.1f us" % (sys.argv[1], N, code, M, e0.elapsed_time(e1) * 20))
else:
    for nk, code in ((512, 1), (256, 1), (256, 0)):
        for tag, v in (("small", "1000000"), ("big", "0")):
            subprocess.run([sys.executable, __file__, tag, str(nk), str(code)], env=dict(os.environ, RN_GEMM_SMALL_BELOW=v))
