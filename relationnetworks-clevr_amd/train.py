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
def load_tensor_data(batch, device, invert_questions=True, out=None):
    """utils.py:133-150: optionally reverse the token order of every question (padding zeros end up
    FIRST, quirk C5), move to the device, turn the 1-based answer indices (B,1) into 0-based (B,).
    out = (img, qst, label) device tensors (DataParallelTrainer.input_buffers): the host -> device copies land THERE when the
    batch has their shapes and dtypes -- the step then needs no device-to-device hand-off copy -- and `out` is returned."""
    qst = batch["question"]
    if invert_questions:
        qst = torch.flip(qst, dims=[1])
    img, ans = batch["image"], batch["answer"]
    if out is not None and all(torch.is_tensor(t) for t in out):
        o_img, o_qst, o_lab = out
        lab = ans.reshape(-1) - 1                               # (on the host: B integers)
        if (img.shape == o_img.shape and img.dtype == o_img.dtype and qst.shape == o_qst.shape and qst.dtype == o_qst.dtype
                and lab.shape == o_lab.shape and lab.dtype == o_lab.dtype):
            o_img.copy_(img, non_blocking=True); o_qst.copy_(qst, non_blocking=True); o_lab.copy_(lab, non_blocking=True)
            return o_img, o_qst, o_lab
    img = img.to(device, non_blocking=True)
    qst = qst.to(device, non_blocking=True)
    label = (ans.to(device, non_blocking=True) - 1).reshape(-1)
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


class SyntheticRelationalTask(SyntheticClevr):
    """A LEARNABLE stand-in for CLEVR (no dataset in this environment): the answer is a function of the image AND the question,
    so a training run has something to converge to -- what the convergence checks of the arithmetic modes train on
    (tests/test_convergence.py, bench.py --convergence).  Image: uniform noise in [0, 0.3) with ONE 40 x 40 square of one of
    three colours (a single channel at 0.9) centred in one of the four quadrants.  Question: 20 tokens, all 3 except the LAST
    one, which asks 1 = "which colour?" or 2 = "which quadrant?" (after the reference's question reversal, utils.py:138-141, that
    token comes first -- so the batches put it first and load_tensor_data flips it to the end).  Answer (1-based like the
    reference's, utils.py:149): colour -> 1..3, quadrant -> 4..7.  Same collate format as SyntheticClevr."""

    def __iter__(self):
        g = torch.Generator().manual_seed(self.seed)
        hw = self.hw
        side, q2 = (5 * hw) // 16, hw // 2
        for _ in range(len(self)):
            B = self.bs
            img = torch.rand(B, 3, hw, hw, generator=g) * 0.3
            colour = torch.randint(0, 3, (B,), generator=g)
            quad = torch.randint(0, 4, (B,), generator=g)
            kind = torch.randint(1, 3, (B,), generator=g)
            for s_ in range(B):
                cy, cx = q2 // 2 + (int(quad[s_]) // 2) * q2, q2 // 2 + (int(quad[s_]) % 2) * q2
                img[s_, int(colour[s_]), cy - side // 2:cy + side // 2, cx - side // 2:cx + side // 2] = 0.9
            qst = torch.full((B, self.max_len), 3, dtype=torch.int64)
            qst[:, 0] = kind
            ans = torch.where(kind == 1, colour + 1, quad + 4).reshape(B, 1)
            yield {"image": img, "question": qst, "answer": ans}


class SyntheticPairRelationTask(SyntheticClevr):
    """A RELATIONAL learnable task (the one-square task above needs no pair reasoning): three 20 x 20 squares,
    one per colour channel, in three distinct cells of a 4 x 4 grid over noise in [0, 0.3).  Question = (kind, colour c) in the
    first two tokens (the rest is filler 7; load_tensor_data's reversal moves them to the end like the reference's, utils.py:138-141):
      kind 1: in which COLUMN is the c square?                               -> answers 1..4   (one object)
      kind 2: which colour has the square CLOSEST to the c square?           -> answers 5..7   (a comparison over pairs; ties are
              excluded by construction: squared grid distances to the two others differ)
      kind 3: how many OTHER squares share the c square's ROW?               -> answers 8..10  (a count over pairs)
    1-based answers like the reference's (utils.py:149)."""

    def __iter__(self):
        g = torch.Generator().manual_seed(self.seed)
        hw = self.hw
        cell, side = hw // 4, (5 * hw) // 32
        for _ in range(len(self)):
            B = self.bs
            img = torch.rand(B, 3, hw, hw, generator=g) * 0.3
            kind = torch.randint(1, 4, (B,), generator=g)
            col = torch.randint(0, 3, (B,), generator=g)
            ans = torch.zeros(B, 1, dtype=torch.int64)
            for s_ in range(B):
                while True:                                             # three distinct cells, no tie for "closest"
                    cells = torch.randperm(16, generator=g)[:3].tolist()
                    pos = [(c_ // 4, c_ % 4) for c_ in cells]
                    c0 = int(col[s_])
                    o = [i for i in range(3) if i != c0]
                    d = [(pos[c0][0] - pos[i][0]) ** 2 + (pos[c0][1] - pos[i][1]) ** 2 for i in o]
                    if d[0] != d[1]:
                        break
                for ch, (r_, c_) in enumerate(pos):
                    y0, x0 = r_ * cell + (cell - side) // 2, c_ * cell + (cell - side) // 2
                    img[s_, ch, y0:y0 + side, x0:x0 + side] = 0.9
                k_ = int(kind[s_])
                if k_ == 1:
                    ans[s_, 0] = 1 + pos[c0][1]
                elif k_ == 2:
                    ans[s_, 0] = 5 + (o[0] if d[0] < d[1] else o[1])
                else:
                    ans[s_, 0] = 8 + sum(1 for i in o if pos[i][0] == pos[c0][0])
            qst = torch.full((B, self.max_len), 7, dtype=torch.int64)
            qst[:, 0] = kind
            qst[:, 1] = 4 + col
            yield {"image": img, "question": qst, "answer": ans}


class PairRelationTaskOnDevice:
    """SyntheticPairRelationTask's task with the batches made ON THE DEVICE (the CPU iterator above spends ~40 ms a batch in
    per-sample Python, 50 x a training step: a multi-seed study would be all data generation).  Same images (noise in [0, 0.3),
    three 20 x 20 squares at 0.9, one per colour channel, in three distinct cells of a 4 x 4 grid, no tie for "closest"), same
    questions and answers; the cell draws are numpy's (vectorised rejection of ties), the noise is the device generator's --
    so the stream differs from the CPU class's, and is the same for every arithmetic mode at a given seed.  Yields what
    load_tensor_data returns: (img, reversed question, 0-based label) device tensors."""

    def __init__(self, n_batches, batch_size, seed=0, device="cuda", hw=128, max_len=20):
        import numpy as np
        self.nb, self.bs, self.seed, self.device, self.hw, self.max_len = n_batches, batch_size, seed, torch.device(device), hw, max_len
        rng = np.random.default_rng(seed)
        n = n_batches * batch_size
        kind = rng.integers(1, 4, n)
        col = rng.integers(0, 3, n)
        cells = np.zeros((n, 3), dtype=np.int64)
        todo = np.arange(n)
        while todo.size:                                                # three distinct cells; redraw the ties of "closest"
            c = np.argsort(rng.random((todo.size, 16)), axis=1)[:, :3]
            cells[todo] = c
            r_, c_ = cells[todo] // 4, cells[todo] % 4
            d = (r_ - r_[np.arange(todo.size), col[todo]][:, None]) ** 2 + (c_ - c_[np.arange(todo.size), col[todo]][:, None]) ** 2
            d[np.arange(todo.size), col[todo]] = -1                     # the asked square itself
            ds = np.sort(d, axis=1)                                     # (-1, nearer, farther)
            todo = todo[ds[:, 1] == ds[:, 2]]
        row, colm = cells // 4, cells % 4
        ar = np.arange(n)
        d = (row - row[ar, col][:, None]) ** 2 + (colm - colm[ar, col][:, None]) ** 2
        d[ar, col] = 1 << 20
        closest = np.argmin(d, axis=1)
        same_row = (row == row[ar, col][:, None]).sum(axis=1) - 1
        ans = np.where(kind == 1, colm[ar, col], np.where(kind == 2, 4 + closest, 7 + same_row))       # 0-based
        self.kind, self.col, self.cells, self.ans = kind, col, cells, ans

    def __len__(self):
        return self.nb

    def __iter__(self):
        dev, hw, B = self.device, self.hw, self.bs
        g = torch.Generator(device=dev).manual_seed(self.seed)
        cell, side = hw // 4, (5 * hw) // 32
        ax = torch.arange(hw, device=dev)
        for t in range(self.nb):
            sl = slice(t * B, (t + 1) * B)
            img = torch.rand(B, 3, hw, hw, generator=g, device=dev) * 0.3
            cells = torch.from_numpy(self.cells[sl]).to(dev)
            y0 = (cells // 4) * cell + (cell - side) // 2                                  # (B, 3)
            x0 = (cells % 4) * cell + (cell - side) // 2
            my = (ax[None, None, :] >= y0[:, :, None]) & (ax[None, None, :] < y0[:, :, None] + side)        # (B, 3, hw)
            mx = (ax[None, None, :] >= x0[:, :, None]) & (ax[None, None, :] < x0[:, :, None] + side)
            img = torch.where(my[:, :, :, None] & mx[:, :, None, :], torch.full_like(img, 0.9), img)
            qst = torch.full((B, self.max_len), 7, dtype=torch.int64)
            qst[:, 0] = torch.from_numpy(self.kind[sl])
            qst[:, 1] = torch.from_numpy(4 + self.col[sl])
            yield img, torch.flip(qst, dims=[1]).to(dev), torch.from_numpy(self.ans[sl]).to(dev)


TASKS = {"square": SyntheticRelationalTask, "pairs": SyntheticPairRelationTask}


def convergence_run(precision, steps=400, batch=64, lr=1e-3, seed=0, device="cuda", h8=None, eval_batches=4, log_every=25,
                    model_name="original-fp", use_graph=True, task="square"):
    """Train `model_name` for `steps` Adam steps (clip 50, weight decay 1e-4: train.py:45-46,330) on SyntheticRelationalTask in the
    arithmetic mode `precision`, same seeds whatever the mode -> {"loss": [mean loss per `log_every` steps], "final_loss",
    "accuracy" (held-out batches, eval mode), "copy_guard": the trainer's e4m3 guard log}.  The run the convergence test and
    bench.py --convergence compare across modes."""
    import contextlib
    import sys
    _opt = sys.modules[type(dp.OPT).__module__]                  # the options module dp imported (package or flat import alike)
    hyp = dict(json.load(open(os.path.join(_HERE, "config.json")))["hyperparams"][model_name], precision=precision)

    class _A:
        qdict_size, adict_size = 82, 28
    torch.manual_seed(seed)
    with contextlib.redirect_stdout(open(os.devnull, "w")):
        model = RN(_A, hyp)
    model.cuda(torch.device(device)) if str(device).startswith("cuda") else None
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=lr, weight_decay=1e-4)
    ctx = _opt.override(h8=h8) if h8 is not None else contextlib.nullcontext()
    with ctx:
        tr = dp.DataParallelTrainer(model, opt, clip_norm=50.0, use_graph=use_graph)
        on_device = task == "pairs_dev"                        # batches made on the device: already (img, reversed qst, 0-based label)
        data = (PairRelationTaskOnDevice(steps + eval_batches, batch, seed=seed + 1, device=device) if on_device
                else TASKS[task]((steps + eval_batches) * batch, batch, seed=seed + 1))
        it = iter(data)
        load = (lambda b_, d_: b_) if on_device else load_tensor_data
        curve, acc_l = [], []
        for st in range(steps):
            img, qst, lab = load(next(it), device)
            acc_l.append(tr.step(img, qst, lab).detach().clone())
            if (st + 1) % log_every == 0:
                curve.append(float(torch.stack(acc_l).mean()))
                acc_l = []
        model.eval()
        hit = tot = 0
        with torch.no_grad():
            for _ in range(eval_batches):
                img, qst, lab = load(next(it), device)
                hit += int((model(img, qst).argmax(1) == lab).sum())
                tot += lab.numel()
    return {"precision": precision, "task": task, "seed": seed, "h8": _opt.OPT.h8 if h8 is None else h8, "steps": steps, "batch": batch, "lr": lr, "loss": curve,
            "final_loss": curve[-1], "accuracy": hit / tot, "copy_guard": tr.copy_guard_log}


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
    bufs = None
    for batch_idx, batch in enumerate(loader):
        img, qst, label = load_tensor_data(batch, device, invert_questions, out=bufs)
        if bufs is None and getattr(trainer, "use_graph", False) and hasattr(trainer, "input_buffers"):
            # from the second batch on the host -> device copies land in the step graph's own input tensors (no hand-off copy)
            bufs = trainer.input_buffers(img, qst, label)
            if bufs[0] is not img:
                for d_, s_ in zip(bufs, (img, qst, label)):
                    d_.copy_(s_)
                img, qst, label = bufs
        loss = trainer.step(img, qst, label).detach()
        running = loss.clone() if running is None else running + loss
        n_run += 1
        if batch_idx % log_interval == 0:
            avg = float(running) / n_run
            bs = label.shape[0]
            processed, total = batch_idx * bs, max(n_batches * bs, 1)
            log("Train Epoch: {} [{}/{} ({:.0%})] Train loss: {}".format(epoch, processed, total, processed / total, avg))
            running, n_run = None, 0


# CLEVR's answer vocabulary by question family (dataset facts; the reference keeps the same table in utils.py:9-16)
ANSWER_CLASSES = {
    "number": [str(i) for i in range(11)],
    "material": ["rubber", "metal"],
    "color": ["cyan", "blue", "yellow", "purple", "red", "green", "gray", "brown"],
    "shape": ["sphere", "cube", "cylinder"],
    "size": ["large", "small"],
    "exist": ["yes", "no"],
}


def default_dictionaries(adict_size=28, qdict_size=82):
    """A (quest_to_ix, answ_to_ix, answ_ix_to_class) triple shaped like utils.build_dictionaries' result (utils.py:18-59)
    for runs without the dataset: 1-based indices, the 28 CLEVR answers in ANSWER_CLASSES order (any further index gets
    the class 'other')."""
    answers = [(a, c) for c, vals in ANSWER_CLASSES.items() for a in vals]
    answ_to_ix, ix_to_class = {}, {}
    for i in range(adict_size):
        a, c = answers[i] if i < len(answers) else ("answer%d" % i, "other")
        answ_to_ix[a] = i + 1
        ix_to_class[i + 1] = c
    quest_to_ix = {"w%d" % i: i for i in range(1, qdict_size + 1)}
    return quest_to_ix, answ_to_ix, ix_to_class


class EvalBookkeeper:
    """The bookkeeping of the reference's test() (train.py:69-158) with the per-sample Python loops replaced by device
    tensors: per batch ONE bincount into an (A, A) [label, prediction] confusion matrix plus the
    prediction / label vectors kept on the device; everything the reference reports is derived at the end --
      class_corrects / class_invalids / class_total_samples per answer class (train.py:71-80, 107-115; an "invalid" is
      a prediction whose class differs from the label's class), the global counters (train.py:121-125), the two
      per-sample confusion lists in sample order (train.py:116-118) and the label strings (train.py:92-94).
    Deviation, on purpose: the reference orders the confusion axes by Python's per-process-randomised hash() of the class
    name (train.py:86), here classes are ordered by name -- 'number' answers numerically, as there -- so the order is
    reproducible; and 'global_accuracy' is the global accuracy (the reference's variable has been overwritten by the last
    class's accuracy when it is dumped, train.py:139-157)."""

    def __init__(self, dictionaries, adict_size, device):
        _, answ_to_ix, ix_to_class = dictionaries
        self.A, self.device = adict_size, device
        inv = {v: k for k, v in answ_to_ix.items()}                                 # 1-based index -> answer string
        self.class_of = [ix_to_class.get(a + 1, "other") for a in range(adict_size)]   # 0-based answer -> class name
        self.class_names = list(dict.fromkeys(ix_to_class[k] for k in ix_to_class))  # insertion order, like the reference's dicts
        if "other" in self.class_of and "other" not in self.class_names:
            self.class_names.append("other")

        def order(a):                       # position of 0-based answer a on the confusion axes
            c = self.class_of[a]
            return (c, int(inv[a + 1]) if c == "number" else a)
        self.sorted_classes = sorted(range(adict_size), key=order)                    # axis position -> 0-based answer
        self.sorted_labels = [inv.get(a + 1, str(a)) for a in self.sorted_classes]
        pos = [0] * adict_size
        for p_, a in enumerate(self.sorted_classes):
            pos[a] = p_
        self.axis_pos = torch.tensor(pos, dtype=torch.long, device=device)            # 0-based answer -> axis position
        self.conf = torch.zeros(adict_size, adict_size, dtype=torch.long, device=device)   # [label, prediction]
        self.loss_sum = torch.zeros((), dtype=torch.float32, device=device)
        self.preds, self.labels = [], []
        self.n_batches = 0

    @torch.no_grad()
    def update(self, log_probs, label):
        """One batch: no host synchronisation."""
        pred = log_probs.argmax(1)
        self.conf += torch.bincount(label * self.A + pred, minlength=self.A * self.A).view(self.A, self.A)
        self.loss_sum += torch.nn.functional.nll_loss(log_probs, label)               # mean per batch (train.py:104, 127)
        self.preds.append(pred)
        self.labels.append(label)
        self.n_batches += 1

    def finalize(self):
        """-> dict: the seven test.pickle keys (train.py:149-157) + 'corrects', 'invalids', 'n_samples', 'avg_loss',
        'confusion' (the (A, A) matrix).  One device -> host transfer."""
        conf = self.conf.cpu()
        n = int(conf.sum())
        same_class = torch.tensor([[self.class_of[l] == self.class_of[p] for p in range(self.A)] for l in range(self.A)])
        corrects_per_label = conf.diag()
        invalid_per_label = (conf * (~same_class)).sum(1)
        total_per_label = conf.sum(1)
        cc, ci, cn = {}, {}, {}
        for c in self.class_names:
            rows = torch.tensor([self.class_of[a] == c for a in range(self.A)])
            cc[c] = int(corrects_per_label[rows].sum())
            ci[c] = int(invalid_per_label[rows].sum())
            cn[c] = int(total_per_label[rows].sum())
        corrects, invalids = int(corrects_per_label.sum()), int(invalid_per_label.sum())
        if self.preds:
            pred = self.axis_pos[torch.cat(self.preds)].cpu().tolist()
            targ = self.axis_pos[torch.cat(self.labels)].cpu().tolist()
        else:
            pred, targ = [], []
        return {"class_corrects": cc, "class_invalids": ci, "class_total_samples": cn,
                "confusion_matrix_target": targ, "confusion_matrix_pred": pred, "confusion_matrix_labels": self.sorted_labels,
                "global_accuracy": corrects / max(n, 1),
                "corrects": corrects, "invalids": invalids, "n_samples": n,
                "avg_loss": float(self.loss_sum) / max(self.n_batches, 1), "confusion": conf}


PICKLE_KEYS = ("class_corrects", "class_invalids", "class_total_samples", "confusion_matrix_target", "confusion_matrix_pred",
               "confusion_matrix_labels", "global_accuracy")


def format_test_log(epoch, res):
    """The lines the reference prints after an evaluation pass (train.py:138-145); plot.py:59,80 parse
    'Accuracy = <x>%' / 'Invalids = <x>%' / 'Test loss = <x>' and plot.py:63 '<class> -- acc: <x>%'."""
    n = max(res["n_samples"], 1)
    lines = ["Test Epoch {}: Accuracy = {:.2%} ({:g}/{}); Invalids = {:.2%} ({:g}/{}); Test loss = {}".format(
        epoch, res["corrects"] / n, float(res["corrects"]), res["n_samples"], res["invalids"] / n, float(res["invalids"]),
        res["n_samples"], res["avg_loss"])]
    for c, tot in res["class_total_samples"].items():
        acc = res["class_corrects"][c] / tot if tot else 0
        inv = res["class_invalids"][c] / tot if tot else 0
        lines.append("{} -- acc: {:.2%} ({}/{}); invalid: {:.2%} ({}/{})".format(c, acc, res["class_corrects"][c], tot, inv,
                                                                                  res["class_invalids"][c], tot))
    return lines


@torch.no_grad()
def test_epoch(loader, model, epoch, device, adict_size, invert_questions=True, log=print, dictionaries=None,
               results_dir=None):
    """The reference's test() (train.py:66-159): eval-mode forward over the loader, accuracy / invalids / per-class
    figures, the log lines plot.py parses and -- with results_dir -- `test.pickle` with the reference's seven keys.
    Returns (average per-batch loss, result dict)."""
    import pickle
    model.eval()
    book = EvalBookkeeper(dictionaries or default_dictionaries(adict_size), adict_size, device)
    for batch in loader:
        img, qst, label = load_tensor_data(batch, device, invert_questions)
        book.update(model(img, qst), label)
    res = book.finalize()
    if str(device).startswith("cuda"):
        # (an evaluation never goes through the trainer's guard -- an unanswered hand-off inside the feature-split f_phi
        # launch leaves an error word and garbage log-probs; finalize() has synchronised)
        st = dp.RF.H.f_phi_split_status(torch.device(device))
        if st:
            raise RuntimeError("rn_f_phi_split: a hand-off inside the launch was not answered during the evaluation (stage %d): "
                               "the figures above are invalid -- re-run with RN_NO_FPHI_SPLIT=1" % (st - 1))
    for line in format_test_log(epoch, res):
        log(line)
    if results_dir is not None:
        os.makedirs(results_dir, exist_ok=True)
        with open(os.path.join(results_dir, "test.pickle"), "wb") as f:
            pickle.dump({k: res[k] for k in PICKLE_KEYS}, f)
    return res["avg_loss"], res


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
    ap.add_argument("--precision", default=None, choices=[None, "auto", "f16s", "bf16", "fp32", "bf16x3"],
                    help='g_theta arithmetic (DESIGN.md section 2); default "auto" = f16s where the fused chain applies')
    ap.add_argument("--test-results-dir", default="./test_results", help="where test.pickle goes (train.py:232)")
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
        test_epoch(test, model, epoch, device, args.adict_size, not args.no_invert_questions, log,
                   results_dir=args.test_results_dir if rank == 0 else None)
        if rank == 0:
            save_checkpoint(model, os.path.join(args.model_dir, args.model), epoch)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
