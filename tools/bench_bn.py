import sys, os, time
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT)
import torch, torch.nn.functional as F
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/n*1e6
for shape in [(64,24,64,64),(64,24,32,32),(64,24,16,16),(64,24,8,8)]:
    x=torch.randn(*shape,device='cuda',requires_grad=True); w=torch.ones(24,device='cuda',requires_grad=True); b=torch.zeros(24,device='cuda',requires_grad=True)
    rm=torch.zeros(24,device='cuda'); rv=torch.ones(24,device='cuda')
    for en in (True,False):
        def f():
            y=torch.batch_norm(x,w,b,rm,rv,True,0.1,1e-5,en); y.sum().backward()
        print(shape,"miopen" if en else "native","fwd+bwd %.1f us"%timeit(f))
