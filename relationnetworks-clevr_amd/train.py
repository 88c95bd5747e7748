"""Training / evaluation driver for the MI355X Relation Network.

A modern-PyTorch counterpart of the reference's loop (`/root/reference/train.py:29-63, 202-364`): it
reproduces the loop *semantics* -- question reversal and 1-based -> 0-based labels
(utils.py:133-150), mean NLL loss, global-norm clip at 50, Adam with coupled weight decay 1e-4, the
"slow start" learning-rate schedule (x2 every 20 epochs from 5e-6, capped at 5e-4), one checkpoint
per epoch named RN_epoch_XX.pth, `module.`-prefix tolerant resume, and the log lines that
plot.py's regexes parse -- on top of the data-parallel trainer in dp.py (one process per GPU, one
RCCL all-reduce per step, hipGraph replay of forward+backward).

The CLEVR dataset connectors are out of scope (no dataset in this environment); any iterable that
yields dicts {'image', 'question', 'answer'} shaped like the reference's collate output works, and
`SyntheticClevr` provides such batches for smoke runs and benchmarks:

    torchrun --nproc-per-node 8 -m relationnetworks_clevr_amd.train --model original-fp --synthetic 6400
"""
from __future__ import annotations

import argparse
import json
import os
import re
import time

import torch
import torch.distributed as dist

try:
    from . import dp
    from .model import RN
except ImportError:                       # flat import, like the reference
    import dp                             # type: ignore
    from model import RN                  # type: ignore

_HERE = os.path.dirname(os.path.abspath(__file__))


# ------------------------------------------------------------------------------------------ data
def load_tensor_data(batch, device, invert_questions=True):
    """utils.py:133-150: optionally reverse the token order of every question (padding zeros end up
    FIRST, quirk C5), move to the device, turn the 1-based answer indices (B,1) into 0-based (B,)."""
    qst = batch["question"]
    if invert_questions:
        qst = torch.flip(qst, dims=[1])
    img = batch["image"].to(device, non_blocking=True)
    qst = qst.to(device, non_blocking=True)
    label = (batch["answer"].to(device, non_blocking=True) - 1).reshape(-1)
    return img, qst, label


class SyntheticClevr:
    """Batches with the shapes / value ranges of the reference's collate output: images in [0,1)
    (ToTensor), right-padded int64 questions with 0 = padding, answers 1-based (B,1)."""

    def __init__(self, n_samples, batch_size, state_description=False, qdict_size=82, adict_size=28, hw=128, seed=0,
                 max_len=20):
        self.n, self.bs, self.sd = n_samples, batch_size, state_description
        self.qd, self.ad, self.hw, self.seed, self.max_len = qdict_size, adict_size, hw, seed, max_len

    def __len__(self):
        return self.n // self.bs

    def __iter__(self):
        g = torch.Generator().manual_seed(self.seed)
        for _ in range(len(self)):
            B = self.bs
            if self.sd:
                img = torch.randn(B, 12, 7, generator=g)
                n_obj = torch.randint(3, 11, (B,), generator=g)
                img = img * (torch.arange(12)[None, :, None] < n_obj[:, None, None])        # zero-padded objects
            else:
                img = torch.rand(B, 3, self.hw, self.hw, generator=g)
            qst = torch.randint(1, self.qd + 1, (B, self.max_len), generator=g)
            lens = torch.randint(5, self.max_len + 1, (B,), generator=g)
            qst = qst * (torch.arange(self.max_len)[None, :] < lens[:, None])
            yield {"image": img, "question": qst, "answer": torch.randint(1, self.ad + 1, (B, 1), generator=g)}


# -------------------------------------------------------------------------------------- schedule
def lr_for_epoch(epoch, base_lr=5e-6, lr_max=5e-4, lr_gamma=2.0, lr_step=20):
    """Learning rate in effect DURING `epoch` (1-based) of a run started at epoch 1, reproducing the
    reference's StepLR usage literally (train.py:330-333, 349-350): `scheduler.last_epoch` starts at 1
    and is stepped once per epoch while the current rate is still below lr_max, so the rate is
    base_lr * gamma ** ((epoch + 1) // lr_step) -- it doubles at epochs 19, 39, ... -- and it freezes at
    the first value >= lr_max (6.4e-4 with the defaults, i.e. slightly ABOVE the nominal 5e-4 cap)."""
    lr, last_epoch = base_lr, 1
    for _e in range(1, epoch + 1):
        if lr_max < 0 or lr < lr_max:
            last_epoch += 1
            lr = base_lr * lr_gamma ** (last_epoch // lr_step)
    return lr


def batch_size_for_epoch(epoch, bs, bs_max=-1, bs_gamma=1.0, bs_step=20):
    """Batch size in effect during `epoch` (train.py:337-341): re-evaluated at the first epoch and at
    every multiple of bs_step as floor(bs * bs_gamma ** (epoch // bs_step)), clipped to bs_max when
    bs_max > 0 (and frozen once it has reached bs_max)."""
    cur = bs
    for e in range(1, epoch + 1):
        if (bs_max < 0 or cur < bs_max) and (e % bs_step == 0 or e == 1):
            cur = int(bs * bs_gamma ** (e // bs_step))
            if bs_max > 0 and cur > bs_max:
                cur = bs_max
    return cur


# ------------------------------------------------------------------------------------ checkpoints
def strip_module_prefix(state):
    """Checkpoints saved from nn.DataParallel carry a 'module.' prefix (train.py:264-274)."""
    return {(k[len("module."):] if k.startswith("module.") else k): v for k, v in state.items()}


def load_checkpoint(model, path, conv_only=False):
    """--resume / --conv-transfer-learn (train.py:264-312): strict load, or only the conv.* tensors."""
    state = strip_module_prefix(torch.load(path, map_location="cpu", weights_only=False))
    if conv_only:
        state = {k[len("conv."):]: v for k, v in state.items() if k.startswith("conv.")}
        res = model.conv.load_state_dict(state, strict=False)
    else:
        res = model.load_state_dict(state, strict=False)
    bad = [k for k in res.missing_keys if not k.endswith("num_batches_tracked")] + list(res.unexpected_keys)
    if bad:
        raise RuntimeError("checkpoint %s does not match the model: %s" % (path, bad))
    m = re.search(r"epoch_(\d+)", os.path.basename(path))
    return int(m.group(1)) if m else 0


def save_checkpoint(model, model_dir, epoch):
    os.makedirs(model_dir, exist_ok=True)
    path = os.path.join(model_dir, "RN_epoch_{:02d}.pth".format(epoch))
    torch.save(model.state_dict(), path)
    return path


# ------------------------------------------------------------------------------------------ loops
def train_epoch(loader, trainer, epoch, device, log_interval=10, invert_questions=True, log=print):
    """One epoch of train.py:29-63.  The loss is read back (a host sync) only every log_interval
    batches instead of every batch."""
    trainer.model.train()
    running, n_run = None, 0
    n_batches = len(loader) if hasattr(loader, "__len__") else 0
    for batch_idx, batch in enumerate(loader):
        img, qst, label = load_tensor_data(batch, device, invert_questions)
        loss = trainer.step(img, qst, label).detach()
        running = loss.clone() if running is None else running + loss
        n_run += 1
        if batch_idx % log_interval == 0:
            avg = float(running) / n_run
            bs = label.shape[0]
            processed, total = batch_idx * bs, max(n_batches * bs, 1)
            log("Train Epoch: {} [{}/{} ({:.0%})] Train loss: {}".format(epoch, processed, total, processed / total, avg))
            running, n_run = None, 0


@torch.no_grad()
def test_epoch(loader, model, epoch, device, adict_size, invert_questions=True, log=print):
    """Accuracy + per-answer confusion counts kept ON the device (row N4): one bincount per batch
    instead of the reference's per-sample Python loops (train.py:98-127)."""
    model.eval()
    conf = torch.zeros(adict_size, adict_size, dtype=torch.long, device=device)       # [label, prediction]
    loss_sum = torch.zeros((), device=device)
    n = 0
    for batch in loader:
        img, qst, label = load_tensor_data(batch, device, invert_questions)
        out = model(img, qst)
        loss_sum += torch.nn.functional.nll_loss(out, label, reduction="sum")
        pred = out.argmax(1)
        conf += torch.bincount(label * adict_size + pred, minlength=adict_size * adict_size).view(adict_size, adict_size)
        n += label.shape[0]
    correct = int(conf.diag().sum())
    acc = 100.0 * correct / max(n, 1)
    log("Test Epoch {}: Accuracy = {:.3f}% ({}/{}); Test loss = {}".format(epoch, acc, correct, n, float(loss_sum) / max(n, 1)))
    return acc, conf.cpu()


# -------------------------------------------------------------------------------------------- CLI
def build_argparser():
    ap = argparse.ArgumentParser(description="Relation Network training on MI355X (reference: train.py:367-418)")
    ap.add_argument("--model", default="original-fp", help="key into config.json's hyperparams")
    ap.add_argument("--config", default=os.path.join(_HERE, "config.json"))
    ap.add_argument("--batch-size", type=int, default=640, help="GLOBAL batch size (train.py:370)")
    ap.add_argument("--test-batch-size", type=int, default=640)
    ap.add_argument("--epochs", type=int, default=350)
    ap.add_argument("--lr", type=float, default=5e-6)
    ap.add_argument("--lr-max", type=float, default=5e-4)
    ap.add_argument("--lr-gamma", type=float, default=2.0)
    ap.add_argument("--lr-step", type=int, default=20)
    ap.add_argument("--clip-norm", type=float, default=50.0)
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--log-interval", type=int, default=10)
    ap.add_argument("--resume", default=None)
    ap.add_argument("--conv-transfer-learn", default=None)
    ap.add_argument("--no-invert-questions", action="store_true")
    ap.add_argument("--dropout", type=float, default=-1.0)
    ap.add_argument("--question-injection", type=int, default=-1)
    ap.add_argument("--precision", default=None, choices=[None, "bf16", "fp32"])
    ap.add_argument("--model-dir", default="model_checkpoints")
    ap.add_argument("--synthetic", type=int, default=6400, help="samples per epoch of synthetic data (no CLEVR here)")
    ap.add_argument("--qdict-size", type=int, default=82)
    ap.add_argument("--adict-size", type=int, default=28)
    ap.add_argument("--no-graph", action="store_true")
    return ap


def main(argv=None):
    args = build_argparser().parse_args(argv)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=device)
    log = print if rank == 0 else (lambda *a, **k: None)
    with open(args.config) as f:
        hyp = dict(json.load(f)["hyperparams"][args.model])
    if args.dropout >= 0:
        hyp["dropout"] = args.dropout                                   # train.py:207-208
    if args.question_injection >= 0:
        hyp["question_injection_position"] = args.question_injection    # train.py:209-210
    if args.precision:
        hyp["precision"] = args.precision
    torch.manual_seed(args.seed)
    model = RN(args, hyp)
    model.cuda(device)
    start_epoch = 1
    if args.resume:
        start_epoch = load_checkpoint(model, args.resume) + 1
    if args.conv_transfer_learn:
        load_checkpoint(model, args.conv_transfer_learn, conv_only=True)
    lr = lr_for_epoch(start_epoch, args.lr, args.lr_max, args.lr_gamma, args.lr_step)
    try:
        opt = torch.optim.Adam(model.parameters(), lr=lr, weight_decay=1e-4, fused=True)
    except Exception:
        opt = torch.optim.Adam(model.parameters(), lr=lr, weight_decay=1e-4)
    trainer = dp.DataParallelTrainer(model, opt, clip_norm=args.clip_norm or None, use_graph=not args.no_graph)
    per_rank = max(args.batch_size // world, 1)
    for epoch in range(start_epoch, args.epochs + 1):
        lr = lr_for_epoch(epoch, args.lr, args.lr_max, args.lr_gamma, args.lr_step)
        for gparam in opt.param_groups:
            gparam["lr"] = lr
        log("Current learning rate: {}".format(lr))
        data = SyntheticClevr(args.synthetic // world, per_rank, hyp["state_description"], args.qdict_size, args.adict_size,
                              seed=args.seed + 1000 * epoch + rank)
        t0 = time.time()
        train_epoch(data, trainer, epoch, device, args.log_interval, not args.no_invert_questions, log)
        torch.cuda.synchronize()
        log("Epoch {} done in {:.1f} s".format(epoch, time.time() - t0))
        test = SyntheticClevr(min(args.synthetic, 4 * args.test_batch_size) // world, max(args.test_batch_size // world, 1),
                              hyp["state_description"], args.qdict_size, args.adict_size, seed=args.seed + 7)
        test_epoch(test, model, epoch, device, args.adict_size, not args.no_invert_questions, log)
        if rank == 0:
            save_checkpoint(model, os.path.join(args.model_dir, args.model), epoch)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
