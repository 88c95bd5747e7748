"""debug: two ranks on one GPU over gloo -- all-reduced gradient vs the single-process full-batch gradient, per parameter"""
import sys, os, io, contextlib, socket
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, torch.distributed as dist, torch.multiprocessing as mp
import test_dp_gpu as TD

def worker(rank, world, port, cfg, out):
    from relationnetworks_clevr_amd import dp
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    model = TD._model(cfg, seed=3 + rank)
    opt = TD._adam(model)
    tr = dp.DataParallelTrainer(model, opt, clip_norm=TD.CLIP, use_graph=os.environ.get("DBG_GRAPH", "1") == "1")
    x, q, y = TD._data(cfg, 8)
    sh = 8 // world; sl = slice(rank * sh, (rank + 1) * sh)
    rec = []
    orig = tr._fused_opt.step
    def spy(clip, gs=1.0):
        torch.cuda.synchronize()
        rec.append((tr.bucket.flat.clone() * gs).cpu())
        return orig(clip, gs)
    tr._fused_opt.step = spy
    for _ in range(2):
        tr.step(x[sl].contiguous(), q[sl].contiguous(), y[sl].contiguous())
    names = [(n, p.numel()) for n, p in model.named_parameters() if p.requires_grad]
    if rank == 0:
        torch.save({"rec": rec, "names": names}, out)
    if world > 1:
        dist.barrier(); dist.destroy_process_group()

if __name__ == "__main__":
    cfg = sys.argv[1] if len(sys.argv) > 1 else "original-fp"
    mp.spawn(worker, args=(2, TD._free_port(), cfg, "/tmp/dp2.pt"), nprocs=2, join=True)
    mp.spawn(worker, args=(1, TD._free_port(), cfg, "/tmp/dp1.pt"), nprocs=1, join=True)
    a, b = torch.load("/tmp/dp2.pt"), torch.load("/tmp/dp1.pt")
    for step in range(2):
        off = 0; rows = []
        for n, k in a["names"]:
            ga, gb = a["rec"][step][off:off + k], b["rec"][step][off:off + k]; off += k
            rows.append((float((ga - gb).norm() / max(float(gb.norm()), 1e-30)), n, float(gb.norm())))
        rows.sort(reverse=True)
        print("step", step, [("%.2e" % e, n, "%.2e" % nn) for e, n, nn in rows[:6]])
    # ... and the reference run IN THIS (parent) process, as the test does it
    from relationnetworks_clevr_amd import dp
    torch.cuda.set_device(0)
    model = TD._model(cfg, seed=3)
    opt = TD._adam(model)
    tr = dp.DataParallelTrainer(model, opt, clip_norm=TD.CLIP, use_graph=True)
    x, q, y = TD._data(cfg, 8)
    rec = []
    orig = tr._fused_opt.step
    def spy(clip, gs=1.0):
        torch.cuda.synchronize(); rec.append((tr.bucket.flat.clone() * gs).cpu()); return orig(clip, gs)
    tr._fused_opt.step = spy
    for _ in range(2):
        tr.step(x, q, y)
    for step in range(2):
        off = 0; rows = []
        for n, k in a["names"]:
            ga, gb = rec[step][off:off + k], b["rec"][step][off:off + k]; off += k
            rows.append((float((ga - gb).norm() / max(float(gb.norm()), 1e-30)), n, float(gb.norm())))
        rows.sort(reverse=True)
        print("parent vs spawned single, step", step, [("%.2e" % e, n, "%.2e" % nn) for e, n, nn in rows[:6]])
