#!/usr/bin/env python3
"""What the host-driven work around the graph replay costs per step: the normal step (3 input copies + torch's RNG fills + replay),
the step on the static input tensors (no copies), and the bare replay."""
import os, sys, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
import bench
import relationnetworks_clevr_amd as pkg
from relationnetworks_clevr_amd import dp
pkg.rn_hip.load()
hyp = dict(json.load(open(os.path.join(ROOT, "relationnetworks-clevr_amd", "config.json")))["hyperparams"]["original-fp"], precision="auto")
torch.manual_seed(42)
dev = torch.device("cuda", 0)
model = bench.quiet_rn(pkg, hyp); model.cuda(dev); model.train()
opt = torch.optim.Adam(model.parameters(), lr=5e-6, weight_decay=1e-4, fused=True)
tr = dp.DataParallelTrainer(model, opt, clip_norm=50.0, use_graph=True)
img, qst, lab = bench.make_batch(64, dev, 128)
pkg.rn_hip.TIMER.enabled = False
for _ in range(20):
    tr.step(img, qst, lab)


def t(fn, n=300):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


for rep in range(2):
    print("step(img, qst, lab)          %.1f us" % t(lambda: tr.step(img, qst, lab)))
    print("step(static inputs)          %.1f us" % t(lambda: tr.step(*tr._static)))
    print("bare graph.replay()          %.1f us" % t(lambda: tr._graph.replay()))
    g = tr._graph
    if hasattr(g, "raw_cuda_graph"):
        pass
