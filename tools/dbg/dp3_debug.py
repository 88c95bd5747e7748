"""debug: per-step parameters of 2-rank / 1-rank runs vs a CPU replay of clip+Adam on the recorded gradients"""
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
    tr = dp.DataParallelTrainer(model, opt, clip_norm=TD.CLIP, use_graph=True)
    x, q, y = TD._data(cfg, 8)
    sh = 8 // world; sl = slice(rank * sh, (rank + 1) * sh)
    rec, par = [], []
    P = lambda: torch.cat([p.detach().reshape(-1) for p in tr.bucket.params]).cpu()
    par.append(P())
    orig = tr._fused_opt.step
    def spy(clip, gs=1.0):
        torch.cuda.synchronize(); rec.append((tr.bucket.flat.clone() * gs).cpu()); return orig(clip, gs)
    tr._fused_opt.step = spy
    for _ in range(3):
        tr.step(x[sl].contiguous(), q[sl].contiguous(), y[sl].contiguous()); torch.cuda.synchronize(); par.append(P())
    names = [(n, p.numel()) for n, p in model.named_parameters() if p.requires_grad]
    if rank == 0:
        torch.save({"rec": rec, "par": par, "names": names}, out)
    if world > 1:
        dist.barrier(); dist.destroy_process_group()

def cpu_adam(par0, grads, lr=1e-3, eps=1e-1, wd=1e-4, clip=0.5, b1=0.9, b2=0.999):
    p = par0.double().clone(); m = torch.zeros_like(p); v = torch.zeros_like(p); outs = []
    for t, g in enumerate(grads, 1):
        g = g.double(); tot = g.norm(); g = g * min(1.0, clip / (float(tot) + 1e-6))
        gd = g + wd * p; m = b1 * m + (1 - b1) * gd; v = b2 * v + (1 - b2) * gd * gd
        p = p - lr / (1 - b1 ** t) * m / (v.sqrt() / (1 - b2 ** t) ** 0.5 + eps); outs.append(p.clone())
    return outs

if __name__ == "__main__":
    cfg = "original-fp"
    mp.spawn(worker, args=(2, TD._free_port(), cfg, "/tmp/dp2.pt"), nprocs=2, join=True)
    mp.spawn(worker, args=(1, TD._free_port(), cfg, "/tmp/dp1.pt"), nprocs=1, join=True)
    for tag in ("/tmp/dp2.pt", "/tmp/dp1.pt"):
        d = torch.load(tag)
        ref = cpu_adam(d["par"][0], d["rec"])
        for step in range(3):
            off = 0; rows = []
            for n, k in d["names"]:
                up_ref = ref[step][off:off + k] - d["par"][0][off:off + k].double(); up = d["par"][step + 1][off:off + k].double() - d["par"][0][off:off + k].double(); off += k
                rows.append((float((up - up_ref).norm() / max(float(up_ref.norm()), 1e-30)), n))
            rows.sort(reverse=True)
            print(tag, "step", step, "update vs CPU Adam:", [("%.2e" % e, n) for e, n in rows[:3]])
        print(tag, "init equal to other run:", torch.equal(d["par"][0], torch.load("/tmp/dp1.pt")["par"][0]))
    d2, d1 = torch.load("/tmp/dp2.pt"), torch.load("/tmp/dp1.pt")
    off = 0
    for n, k in d1["names"]:
        if "g_layers" in n and "weight" in n:
            u2 = d2["par"][3][off:off + k] - d2["par"][0][off:off + k]; u1 = d1["par"][3][off:off + k] - d1["par"][0][off:off + k]
            print("spawned", n, "update norms 2-rank %.4e 1-rank %.4e rel diff %.2e" % (float(u2.norm()), float(u1.norm()), float((u2 - u1).norm() / u1.norm())))
        off += k
    # the test's own sequence in THIS process
    from relationnetworks_clevr_amd import dp
    torch.cuda.set_device(0)
    model = TD._model(cfg, seed=3)
    init = {k: v.detach().cpu().float().clone() for k, v in model.state_dict().items()}
    flat0 = torch.cat([p.detach().reshape(-1) for p in model.parameters()]).cpu()
    print("parent init equals spawned init:", torch.equal(flat0, d1["par"][0]))
    opt = TD._adam(model)
    tr = dp.DataParallelTrainer(model, opt, clip_norm=TD.CLIP, use_graph=True)
    x, q, y = TD._data(cfg, 8)
    for _ in range(3):
        tr.step(x, q, y)
    torch.cuda.synchronize()
    for k, v in model.state_dict().items():
        if "g_layers" in k and "weight" in k:
            print("parent ", k, "update norm %.4e" % float((v.cpu().float() - init[k]).norm()))
    off = 0
    for n, k in d1["names"]:
        if n == "rl.g_layers.0.weight":
            for step in range(3):
                g2 = d2["rec"][step][off:off + k].view(256, 180); g1 = d1["rec"][step][off:off + k].view(256, 180)
                print("grad step", step, "rel diff %.2e" % float((g2 - g1).norm() / g1.norm()), " cols x_j %.2e x_i %.2e q %.2e" % tuple(
                    float((g2[:, a:b] - g1[:, a:b]).norm() / g1[:, a:b].norm()) for a, b in ((0, 26), (26, 52), (52, 180))),
                    " |g1| by block %.2e %.2e %.2e" % tuple(float(g1[:, a:b].norm()) for a, b in ((0, 26), (26, 52), (52, 180))))
                p2 = d2["par"][step + 1][off:off + k].view(256, 180) - d2["par"][step][off:off + k].view(256, 180); p1 = d1["par"][step + 1][off:off + k].view(256, 180) - d1["par"][step][off:off + k].view(256, 180)
                print("   update step", step, " cols x_j %.2e x_i %.2e q %.2e" % tuple(float((p2[:, a:b] - p1[:, a:b]).norm() / p1[:, a:b].norm()) for a, b in ((0, 26), (26, 52), (52, 180))))
        off += k
