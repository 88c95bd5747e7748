import sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
import numpy as np, torch, gold
from oracle import formula, rn_oracle as O
import relationnetworks_clevr_amd as pkg
H=pkg.rn_hip; H.load()
for tag in ['G-drop','G-fp-small','G-fp196']:
    g=gold.load(tag); meta=g['meta']; hyp,sd,x,q,lab=gold.rl_case(meta)
    hyp=dict(hyp,precision='fp32')
    rl=pkg.RelationalLayer(hyp['rl_in_size'],28,hyp['lstm_hidden'],hyp); rl.load_state_dict({k:torch.from_numpy(v) for k,v in sd.items()}); rl.cuda()
    if 'dropout_mask' in g: rl.train(); rl.forced_dropout_mask=torch.from_numpy(g['dropout_mask']).cuda()
    else: rl.eval()
    xt=torch.from_numpy(x).cuda().requires_grad_(True); qt=torch.from_numpy(q).cuda().requires_grad_(True)
    lp=rl(xt,qt); torch.nn.functional.nll_loss(lp,torch.from_numpy(lab).cuda()).backward(); torch.cuda.synchronize()
    dx=xt.grad.cpu().numpy(); e=np.abs(dx-g['dx'])/np.abs(g['dx']).max()
    print(tag,'max',e.max(),'per-b',e.max((1,2)),'per-chan',np.round(e.max((0,1))*1e4,2))
    idx=np.unravel_index(e.argmax(),e.shape); print('  worst at',idx,'got',dx[idx],'ref',g['dx'][idx])
    jm=e.max((0,2)); print('  per-j (x1e4)',np.round(jm*1e4,1)[:16],'...')
