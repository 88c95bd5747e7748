"""Shared by the chain timing tools: round-robin timing of launch variants under one duty cycle."""
import torch


def spin_cycles(ms):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); torch.cuda._sleep(1000000); e1.record(); torch.cuda.synchronize()
    return int(1000000 * ms / max(e0.elapsed_time(e1), 1e-3))


def time_variants(variants, reps=15, rest_ms=0.8):
    """Median launch duration of every variant {name: (pre, fn)}, measured ROUND-ROBIN with a low-power spin in front of every
    launch: a kernel launched back to back runs at the clock its own power draw leaves, so variants are only comparable under the
    same duty cycle."""
    spin = spin_cycles(rest_ms)
    ts = {k: [] for k in variants}
    for r in range(reps + 2):
        for name, (pre, fn) in variants.items():
            pre()
            torch.cuda._sleep(spin)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record()
            torch.cuda.synchronize()
            if r >= 2:
                ts[name].append(e0.elapsed_time(e1) * 1e3)
    return {k: sorted(v)[len(v) // 2] for k, v in ts.items()}
