#!/usr/bin/env python3
"""Durations of the latency-bound kernels that follow the backward chain (partial sums -> Rj/Ri/Rq, dx / dq, the layer-0 weight
gradient) ALONE and BESIDE the three-job weight-gradient launch on another stream -- the situation in the captured step."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
import relationnetworks_clevr_amd as pkg
H = pkg.rn_hip; H.load()
B, n, k, Q, G = 64, 64, 26, 128, 256
M = B * n * n; kt = 2 * k + Q
dev = "cuda"
x = torch.randn(B, n, k, device=dev); q = torch.randn(B, Q, device=dev)
W0 = torch.randn(G, kt, device=dev) * 0.05
Rj = torch.randn(B * n, G, device=dev); Ri = torch.randn(B * n, G, device=dev); Rq = torch.randn(B, G, device=dev)
dx = torch.empty(B, n, k, device=dev); dq = torch.empty(B, Q, device=dev); dW0 = torch.empty(G, kt, device=dev); db0 = torch.empty(G, device=dev)
tpu = H.g_chain_bwd_rr_red_tpu(M, n, n)
rjp = torch.randn(M // 256 // tpu, 32, G, device=dev); rip = torch.randn(M // 16, G, device=dev)
dZ = torch.randn(M, G, device=dev).bfloat16(); Hh = torch.randn(M, G, device=dev).abs().to(torch.float8_e4m3fn)
dZb = [H.rows_to_blocked(dZ) for _ in range(2)]; H8 = [H.rows_to_blocked(Hh) for _ in range(3)]
gWs = [torch.empty(G, G, device=dev) for _ in range(3)]; gBs = [torch.empty(G, device=dev) for _ in range(3)]
dxg = torch.randn(B, G, device=dev)
jobs = [(dZb[0], H8[0], gWs[0], gBs[0]), (dZb[1], H8[1], gWs[1], gBs[1]), (None, H8[2], gWs[2], gBs[2])]
side, s2 = torch.cuda.Stream(), torch.cuda.Stream()
rows = [
    ("pair_reduce_parts", lambda: H.pair_reduce_parts(rjp, rip, Rj, Ri, Rq, B, n, G, (n // 8) // tpu)),
    ("pair_dx_dq", lambda: H.pair_dx_dq(Rj, Ri, Rq, W0, dx, dq, B, n, k, Q, G)),
    ("wgrad0_from_reductions", lambda: H.wgrad0_from_reductions(Rj, Ri, Rq, x, q, dW0, db0)),
    ("parts + dx/dq", lambda: (H.pair_reduce_parts(rjp, rip, Rj, Ri, Rq, B, n, G, (n // 8) // tpu), H.pair_dx_dq(Rj, Ri, Rq, W0, dx, dq, B, n, k, Q, G))),
]


def bracket(fn, beside, also=None, reps=15):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        main = torch.cuda.current_stream()
        torch.cuda._sleep(400000)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        start = main.record_event()
        if beside:
            side.wait_event(start)
            with torch.cuda.stream(side):
                H.g_wgrad_blocked(jobs, M, dxg=dxg, rows_per_question=n * n)
        if also is not None:
            s2.wait_event(start)
            with torch.cuda.stream(s2):
                also()
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


for name, fn in rows:
    for _ in range(2):
        fn()
    a = bracket(fn, False); b = bracket(fn, True)
    print("%-28s alone %6.1f us   beside the weight gradient %6.1f us" % (name, a, b))
w0 = rows[2][1]
print("%-28s beside wgrad AND wgrad0 on a third stream %6.1f us" % ("parts + dx/dq", bracket(rows[3][1], True, also=w0)))
print("%-28s beside wgrad0 only %6.1f us" % ("parts + dx/dq", bracket(rows[3][1], False, also=w0)))
