import torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import time_launch
for mb in (100, 443, 1024, 4096):
    t = torch.empty(mb * 1024 * 1024 // 4, device="cuda")
    s = torch.empty_like(t)
    ms = time_launch(lambda: t.fill_(1.0)); print("fill  %5d MB  %.1f us  %.0f GB/s written" % (mb, ms * 1e3, mb * 1.048576 / ms))
    ms = time_launch(lambda: s.copy_(t)); print("copy  %5d MB  %.1f us  %.0f GB/s (read+write)" % (mb, ms * 1e3, 2 * mb * 1.048576 / ms))
    ms = time_launch(lambda: t.sum()); print("sum   %5d MB  %.1f us  %.0f GB/s read" % (mb, ms * 1e3, mb * 1.048576 / ms))
