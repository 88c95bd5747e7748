#!/usr/bin/env python3
"""Time rn_g_wgrad_blocked on the headline shape (M = 64 * 4096): one stored-gradient job (e4m3 / bf16 activation image), the
generated-gradient job, the step's three jobs in one launch; and the general row-major kernel for comparison."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
import relationnetworks_clevr_amd as pkg
H = pkg.rn_hip
if os.environ.get("RN_LIB"):                                # (a variant build: tools/dbg/variant_lib.sh)
    H.LIB_PATH = os.path.abspath(os.environ["RN_LIB"])
H.load()
B, n, G = int(os.environ.get("B", 64)), int(os.environ.get("N_OBJ", 64)), 256
M = B * n * n


def timeit(fn, n=15):
    for _ in range(3):
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


# (the kernel takes row-blocked images; random data: any bytes time the same)
dZ = [((torch.rand(M, G, device="cuda") - 0.5) * 1e-2).bfloat16() for _ in range(3)]
A16 = [(torch.rand(M, G, device="cuda") * 2).bfloat16() for _ in range(3)]
A8 = [a.to(torch.float8_e4m3fn) for a in A16]
mask = torch.randint(0, 256, (H.g_chain_rr_mask_bytes(M),), dtype=torch.uint8, device="cuda")
dxg = torch.rand(B, G, device="cuda") - 0.5
dW = [torch.empty(G, G, device="cuda") for _ in range(3)]; db = [torch.empty(G, device="cuda") for _ in range(3)]
H.relu_gate_image(mask, A8[2], M)                          # the gate job's operand: the gate in the sign bits of its e4m3 image
kw = dict(dxg=dxg, rows_per_question=n * n)
for name, A, z3 in (("e4m3 A", A8, None), ("bf16 A", A16, dZ[2])):
    mb = (M * G * 2 + M * G * A[0].element_size()) / 1e6
    us = timeit(lambda: H.g_wgrad_blocked([(dZ[0], A[0], dW[0], db[0])], M, **kw))
    print("stored dZ, %s      : %7.1f us  (%.0f MB -> %.2f TB/s)" % (name, us, mb, mb / us))
    z3b = z3.element_size() if z3 is not None else 0
    mb = (M * G * z3b + M * G * A[0].element_size()) / 1e6
    us = timeit(lambda: H.g_wgrad_blocked([(z3, A[2], dW[2], db[2])], M, **kw))
    print("last layer (%s), %s: %7.1f us  (%.0f MB -> %.2f TB/s)" % ("gate job" if z3 is None else "stored", name, us, mb, mb / us))
    mb = (2 * M * G * 2 + M * G * z3b + 3 * M * G * A[0].element_size()) / 1e6
    us = timeit(lambda: H.g_wgrad_blocked([(dZ[0], A[0], dW[0], db[0]), (dZ[1], A[1], dW[1], db[1]), (z3, A[2], dW[2], db[2])], M, **kw))
    print("three jobs, %s     : %7.1f us  (%.0f MB -> %.2f TB/s; 1.03e11 flop -> %.3f of 2.5 PF)" % (name, us, mb, mb / us, 3 * 2.0 * M * G * G / (us * 1e-6) / 2.5e15))
us = timeit(lambda: H.g_linear_bwd_wgrad(dZ[0], G, A16[0], G, dW[0], db[0], 0, M, G, G, G))
print("general row-major kernel: %7.1f us" % us)
if os.environ.get("RN_DIAG", "0") == "1":
    for abl, what in ((1, "stream only"), (257, "stream only, every step the same 4 steps' addresses (L2 hits)"), (513, "stream only, no second reader"), (2, "compute only"), (3, "loop + barriers only"), (66, "compute only, no conversions"), (8, "no A frag reads"), (24, "no frag reads at all"), (10, "compute only, no A frag reads"), (26, "compute only, no frag reads")):
        for name, jobs in (("stored, e4m3", [(dZ[0], A8[0], dW[0], db[0])]), ("gate job", [(None, A8[2], dW[2], db[2])]),
                           ("three jobs", [(dZ[0], A8[0], dW[0], db[0]), (dZ[1], A8[1], dW[1], db[1]), (None, A8[2], dW[2], db[2])])):
            us = timeit(lambda: H.g_wgrad_blocked(jobs, M, abl=abl, **kw))
            print("ABL %3d (%s) %s: %7.1f us" % (abl, what, name, us))
