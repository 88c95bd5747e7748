"""Time the fused forward chain kernel alone under ablations (RN_CHAIN_ABLATE bits: 1 = no weight
re-streaming, 2 = no epilogue, 4 = no MFMA; store=0 drops the HBM activation writes)."""
import sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT)
import torch
import relationnetworks_clevr_amd as pkg
H=pkg.rn_hip; H.load()
M,G,K0=262144,256,192
P=torch.randn(M,K0,device='cuda').bfloat16()
Ws=[(torch.randn(G,K0 if l==0 else G,device='cuda')*0.05).bfloat16() for l in range(4)]
bs=[torch.randn(G,device='cuda')*0.1 for _ in range(4)]
Hs=[torch.empty(M,G,dtype=torch.bfloat16,device='cuda') for _ in range(4)]
part=torch.empty(M//H.g_chain_tile(),G,device='cuda')
def t(store,abl,n=10):
    os.environ['RN_CHAIN_ABLATE']=str(abl)
    hs=Hs if store else [None]*4
    for _ in range(3): H.g_chain_fwd(P,K0,Ws,bs,hs,[K0,G,G,G],part,0,M,G)
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): H.g_chain_fwd(P,K0,Ws,bs,hs,[K0,G,G,G],part,0,M,G)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n*1e3
for store in (1,0):
    for abl in (0,):
        print("store=%d abl=%d  %.1f us" % (store,abl,t(store,abl)))
