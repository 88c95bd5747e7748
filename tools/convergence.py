"""Loss curves of the arithmetic modes on a learnable synthetic task (train.convergence_run):
python tools/convergence.py [steps] [lr] [task: square | pairs] [log_every] [model: original-fp | ir-fp ...] [modes: all | short]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import relationnetworks_clevr_amd as pkg
from relationnetworks_clevr_amd import train as T
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
lr = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-3
task = sys.argv[3] if len(sys.argv) > 3 else "square"
log_every = int(sys.argv[4]) if len(sys.argv) > 4 else 25
model_name = sys.argv[5] if len(sys.argv) > 5 else "original-fp"
modes = (("fp32", None), ("auto", True), ("auto", False), ("bf16", True))
if len(sys.argv) > 6 and sys.argv[6] == "short":
    modes = modes[:2]                                     # fp32 beside the default mode
for prec, h8 in modes:
    t0 = time.time()
    r = T.convergence_run(prec, steps=steps, lr=lr, h8=h8, task=task, log_every=log_every, eval_batches=8 if task == "pairs" else 4, model_name=model_name)
    r["model"] = model_name
    r["seconds"] = time.time() - t0
    r["loss"] = [round(v, 4) for v in r["loss"]]
    print(json.dumps(r))
