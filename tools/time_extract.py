"""Time the fused feature-extraction op (rn_extract_features) at B = 64, n = 64 for every hook position; `once` runs each once
(for rocprofv3 --pmc)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import relationnetworks_clevr_amd as pkg
from oracle import formula
H = pkg.rn_hip; H.load()
once = len(sys.argv) > 1 and sys.argv[1] == "once"
for cfg in ("ir-fp", "original-fp"):
    hyp = dict(formula.HYP[cfg]); k, Q = hyp["rl_in_size"] // 2, hyp["lstm_hidden"]
    rl = pkg.RelationalLayer(hyp["rl_in_size"], formula.ADICT, Q, hyp, extraction=True).cuda().eval()
    x = torch.rand(64, 64, k, device="cuda"); q = torch.rand(64, Q, device="cuda")
    for li in range(4):
        rl.extract_features(x, q, li); torch.cuda.synchronize()
        if once:
            continue
        t0 = time.perf_counter()
        for _ in range(10):
            rl.extract_features(x, q, li)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        fl = 2 * 64 * 4096 * (52 * 256 * (li > 0) + 256 * 256 * max(li - 1, 0))
        print("%s layer_idx %d: %.3f ms  (%.1f TFLOP/s fp32, %d images/s)" % (cfg, li, 1e3 * dt, fl / dt / 1e12, 64 / dt))
