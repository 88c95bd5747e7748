import torch, sys
sys.path.insert(0, "/root/repo")
import relationnetworks_clevr_amd as pkg
H = pkg.rn_hip; H.load()
def tm(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (N, C, S) in [(64, 3, 128), (64, 24, 64), (64, 24, 32), (64, 24, 16)]:
    x = torch.randn(N, C, S, S, device="cuda"); dy = torch.randn(N, 24, S // 2, S // 2, device="cuda"); w = torch.randn(24, C, 3, 3, device="cuda")
    dw = torch.empty_like(w)
    a = tm(lambda: H.conv3x3s2_bwd_weight(x, dy, dw))
    b = tm(lambda: torch.ops.aten.convolution_backward(dy, x, w, None, (2, 2), (1, 1), (1, 1), False, (0, 0), 1, [False, True, False]))
    print(f"Cin={C} S={S}: custom {a:7.1f} us   miopen {b:7.1f} us")
