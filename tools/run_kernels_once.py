"""Launch each hot-path kernel a few times at the headline shape (for rocprofv3 --pmc / --kernel-trace runs)."""
import sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT)
import torch
import relationnetworks_clevr_amd as pkg
H=pkg.rn_hip; H.load()
which=sys.argv[1] if len(sys.argv)>1 else "all"
B,n,k,Q,G=64,64,26,128,256
M=B*n*n; K0=192
x=torch.randn(B,n,k,device='cuda'); q=torch.randn(B,Q,device='cuda')
P=torch.empty(M,K0,dtype=torch.bfloat16,device='cuda')
Ws=[(torch.randn(G,K0 if l==0 else G,device='cuda')*0.05).bfloat16() for l in range(4)]
bs=[torch.randn(G,device='cuda')*0.1 for _ in range(4)]
Hs=[torch.empty(M,G,dtype=torch.bfloat16,device='cuda') for _ in range(4)]
T=H.g_chain_tile(); part=torch.empty(M//T,G,device='cuda')
dZ=torch.randn(M,G,device='cuda').bfloat16(); dZ2=torch.empty_like(dZ)
dW=torch.empty(G,G,device='cuda'); db=torch.empty(G,device='cuda')
for it in range(3):
    if which in ("all","build"): H.pair_build_fwd(x,q,P,0,B,n,k,Q,K0)
    if which in ("all","chain"): H.g_chain_fwd(P,K0,Ws,bs,Hs,[K0,G,G,G],part,0,M,G)
    if which in ("all","fwd"): H.g_linear_fwd(Hs[0],G,Ws[1],G,bs[1],Hs[1],G,0,M,G,G)
    if which in ("all","dgrad"): H.g_linear_bwd_dgrad(dZ,G,Ws[1],G,Hs[0],G,dZ2,G,0,M,G,G)
    if which in ("all","wgrad"): H.g_linear_bwd_wgrad(dZ,G,Hs[0],G,dW,db,0,M,G,G,G)
torch.cuda.synchronize()
