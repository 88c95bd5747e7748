"""K1 (pair_build_kernel) alone at the benched shapes, plus a plain fill_ of the same bytes (the streaming-store ceiling).
SHAPES=0,1,2 picks shapes (rocprofv3 runs want one shape per process)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch, relationnetworks_clevr_amd as pkg
from bench import pair_build_k1, time_launch
H = pkg.rn_hip; H.load()
shapes = ((64, 64, 26, 128), (32, 196, 26, 128), (64, 64, 26, 0))
pick = [int(s) for s in os.environ.get("SHAPES", "0,1,2").split(",")]
for idx in pick:
    B, n, k, Q = shapes[idx]
    r = pair_build_k1(H, B, n, k, Q, "cuda")
    P = torch.empty(r["written_incl_padding"] // 2, dtype=torch.bfloat16, device="cuda")
    f = 1e3 * time_launch(lambda: P.fill_(1.0))
    print(B, n, k, Q, "%.1f us  %.0f GB/s  frac %.3f   (fill_ of the %.1f MB written: %.1f us)"
          % (r["us_per_launch"], r["achieved"], r["frac"], r["written_incl_padding"] / 1e6, f))
