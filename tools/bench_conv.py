"""GPU micro-benchmark of the (stock torch/MIOpen) conv + LSTM prologue under a few backend settings."""
import sys, os, time, io, contextlib
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT)
import torch
import relationnetworks_clevr_amd as pkg
def timeit(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/n*1e3
def run(tag, cl=False):
    conv=pkg.ConvInputModel().cuda().train(); text=pkg.QuestionEmbedModel(82).cuda().train()
    img=torch.rand(64,3,128,128,device='cuda'); q=torch.randint(1,83,(64,20),device='cuda')
    if cl: conv=conv.to(memory_format=torch.channels_last); img=img.contiguous(memory_format=torch.channels_last)
    def f():
        conv.zero_grad(); y=conv(img); y.sum().backward()
    def g():
        text.zero_grad(); y=text(q); y.sum().backward()
    print("%-28s conv fwd+bwd %.3f ms   lstm fwd+bwd %.3f ms" % (tag, timeit(f), timeit(g)))
run("default")
run("channels_last", cl=True)
torch.backends.cudnn.benchmark=True; run("cudnn.benchmark"); torch.backends.cudnn.benchmark=False
torch.backends.cudnn.enabled=False; run("miopen disabled"); run("miopen disabled + CL", cl=True)
