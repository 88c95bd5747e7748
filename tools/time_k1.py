import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch, relationnetworks_clevr_amd as pkg
from bench import pair_build_k1
H = pkg.rn_hip; H.load()
for (B, n, k, Q) in ((64, 64, 26, 128), (32, 196, 26, 128), (64, 64, 26, 0)):
    r = pair_build_k1(H, B, n, k, Q, "cuda")
    print(B, n, k, Q, "%.1f us  %.0f GB/s  frac %.3f" % (r["us_per_launch"], r["achieved"], r["frac"]))
