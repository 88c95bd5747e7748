"""Run the relational layer's forward + backward a few times at the headline shape (original-fp, B=64, n=64, M=262144 pair
rows) with eager launches -- the target of the rocprofv3 --pmc / --kernel-trace runs whose summaries live in profiles/.
The kernels launched are exactly the training step's (module default mode unless argv[1] names another precision)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import relationnetworks_clevr_amd as pkg
prec = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] != "all" else "auto"
B, n, k, Q = int(os.environ.get("B", 64)), int(os.environ.get("N_OBJ", 64)), 26, 128
hyp = {"g_layers": [256] * 4, "f_fc1": 256, "f_fc2": 256, "dropout": 0.0, "question_injection_position": 0,
       "rl_in_size": 2 * k, "lstm_hidden": Q, "state_description": False, "precision": prec}
torch.manual_seed(0)
rl = pkg.RelationalLayer(2 * k, 28, Q, hyp).cuda().train()
x = torch.randn(B, n, k, device="cuda", requires_grad=True); q = torch.randn(B, Q, device="cuda", requires_grad=True)
lab = torch.randint(0, 28, (B,), device="cuda")
for it in range(3):
    for p in rl.parameters():
        p.grad = None
    torch.nn.functional.nll_loss(rl(x, q), lab).backward()
torch.cuda.synchronize()
