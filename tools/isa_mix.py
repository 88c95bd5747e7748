#!/usr/bin/env python3
"""Instruction mix of the chain kernels between their first and last MFMA, per MFMA (the chains are ISSUE-bound: every
instruction beside an MFMA costs an issue slot of the SIMD, whatever its type).  Also checks that nothing but the LDS-DMA
sequences touches M0.  usage: tools/isa_mix.py [source.hip] [kernel-name substring ...]"""
import collections, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1].endswith(".hip") else os.path.join(ROOT, "relationnetworks-clevr_amd", "csrc", "rn_chain_rr.hip")
pats = [a for a in sys.argv[1:] if not a.endswith(".hip")] or ["g_chain_rr_f16s_kernelILi4ELb1ELb0ELb1ELb1ELb1ELi0ELb1ELi0ELb0ELb1ELb0E", "g_chain_rr_bwd_kernelILi0ELb1ELb1E"]
flags = [f for f in os.environ.get("ISA_FLAGS", "").split() if f]
with tempfile.TemporaryDirectory() as d:
    out = os.path.join(d, "k.s")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", out, src] + flags, check=True, stderr=subprocess.DEVNULL)
    text = open(out).read().split("\n")
starts = [(i, l.split(":")[0]) for i, l in enumerate(text) if re.match(r"^_Z\w+:", l)]
for pat in pats:
    for n, (i, name) in enumerate(starts):
        if pat not in name:
            continue
        end = starts[n + 1][0] if n + 1 < len(starts) else len(text)
        lines = [l.strip() for l in text[i:end] if l.startswith("\t") and not l.strip().startswith((".", ";"))]
        idx = [j for j, l in enumerate(lines) if l.startswith("v_mfma")]
        body = lines[idx[0]:idx[-1] + 1]
        c = collections.Counter()
        for l in body:
            op = l.split()[0]
            kind = ("mfma" if op.startswith("v_mfma") else "lds" if op.startswith("ds_") else "wait" if op.startswith("s_waitcnt") else
                    "smem" if op.startswith(("s_store", "s_load", "s_dcache")) else "salu" if op.startswith("s_") else
                    "vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else "valu" if op.startswith("v_") else "other")
            c[kind + ":" + ("" if kind in ("mfma", "wait") else op)] += 1
        nm = c["mfma:"]
        groups = collections.Counter()
        for k_, v in c.items():
            groups[k_.split(":")[0]] += v
        m0 = [l for l in lines if re.search(r"\bm0\b", l) and not l.startswith("s_add_i32 m0,")]
        print("%s\n  %d MFMAs, %d instructions between the first and the last = %.2f per MFMA   %s" % (name[:110], nm, len(body), len(body) / nm, {k_: round(v / nm, 2) for k_, v in sorted(groups.items())}))
        print("  scratch / spills:", sum(1 for l in lines if l.startswith("scratch_")), " v_writelane:", sum(1 for l in lines if l.startswith("v_writelane")), " other M0 uses:", len(m0))
        for k_, v in sorted(c.items(), key=lambda kv: -kv[1])[:int(os.environ.get("ISA_TOP", 14))]:
            print("     %-36s %5d  %.3f" % (k_, v, v / nm))
