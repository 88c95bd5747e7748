"""Phase timestamps (s_memtime, 100 MHz ticks? -> reported as raw deltas) of the fused chain kernels."""
import sys, os, ctypes
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT)
import torch
import relationnetworks_clevr_amd as pkg
H=pkg.rn_hip; lib=H.load()
M,G,K0=262144,256,192
P=torch.randn(M,K0,device='cuda').bfloat16()
Ws=[(torch.randn(G,K0 if l==0 else G,device='cuda')*0.05).bfloat16() for l in range(4)]
bs=[torch.randn(G,device='cuda')*0.1 for _ in range(4)]
Hs=[torch.empty(M,G,dtype=torch.bfloat16,device='cuda') for _ in range(4)]
part=torch.empty(M//H.g_chain_tile(),G,device='cuda')
tr=torch.zeros(32*8,dtype=torch.int64,device='cuda')
for _ in range(3): H.g_chain_fwd(P,K0,Ws,bs,Hs,[K0,G,G,G],part,0,M,G)
lib.rn_debug_set_chain_trace.argtypes=[ctypes.c_void_p]; lib.rn_debug_set_chain_trace(tr.data_ptr())
H.g_chain_fwd(P,K0,Ws,bs,Hs,[K0,G,G,G],part,0,M,G); torch.cuda.synchronize()
lib.rn_debug_set_chain_trace(None)
t=tr.cpu().view(8,32)
names=["start","staged"]+sum([["L%dslabs"%l,"L%depi"%l] for l in range(4)],[])+["end"]
for blk in range(0,5):
    row=t[blk]; n=int((row!=0).sum())
    d=[int(row[i+1]-row[i]) for i in range(n-1)]
    print("wg %4d:"%(blk*50), " ".join("%s=%d"%(names[i+1].replace(' ','_'),d[i]) for i in range(len(d))), "total",int(row[n-1]-row[0]))
